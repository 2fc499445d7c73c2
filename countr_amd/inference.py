"""Sliding-window inference around SupervisedMAE.forward.

Reference behaviour: FSC_test_cross(few-shot).py:264-359 and demo_zero.py:41-74 -- images are resized to height 384,
the width is covered by 384-px windows with stride 128 (last window snapped to w-384), overlapping columns are blended
sequentially (previously covered columns: 1/2 old + 1/2 new), count = sum/60, optional 3x3 crop-and-upscale for tiny
exemplars and test-time normalisation by the mean exemplar-box count.  Images narrower than 384 give an all-zero map
(the reference's loop never runs).  Every window is an independent forward, so all windows (of all crops) go through
the engine as ONE batch -- the stitching itself is a few elementwise ops on [384, w].
"""
import torch
import torch.nn.functional as F


def window_starts(width):
    """Start columns produced by the reference loop (FSC_test_cross(few-shot).py:326-349)."""
    starts, start = [], 0
    while start + 383 < width:
        starts.append(start)
        start += 128
        if start + 383 >= width:
            if start == width - 384 + 128:
                break
            start = width - 384
    return starts


def blend_windows(outputs, starts, width, height=384):
    """Sequential blend of per-window densities [n, height, 384] exactly as the reference does it."""
    dm = torch.zeros(height, width, device=outputs.device, dtype=outputs.dtype)
    prev = -1
    for out, start in zip(outputs, starts):
        ov = prev - start + 1
        if ov > 0:
            dm[:, start:prev + 1] = dm[:, start:prev + 1] / 2 + out[:, :ov] / 2
        dm[:, prev + 1:start + 384] = out[:, max(ov, 0):]
        prev = start + 383
    return dm


def _bucket(n, max_batch):
    """Forward batch sizes come from {1, 2, 4, ..., max_batch}: the engine keeps one static plan (buffers + launch lists) per
    batch size, so arbitrary window counts would build arbitrarily many plans.  A chunk is padded up to its bucket."""
    b = 1
    while b < n and b < max_batch:
        b *= 2
    return min(b, max_batch)


def blend_windows_batched(outputs, starts, width, height=384):
    """blend_windows for n images of the SAME width at once: outputs [n, len(starts), height, 384] -> [n, height, width].  The
    same sequential arithmetic per image (previously covered columns: old / 2 + new / 2), one set of launches for all of them."""
    n = outputs.shape[0]
    dm = torch.zeros(n, height, width, device=outputs.device, dtype=outputs.dtype)
    prev = -1
    for k, start in enumerate(starts):
        out = outputs[:, k]
        ov = prev - start + 1
        if ov > 0:
            dm[:, :, start:prev + 1] = dm[:, :, start:prev + 1] / 2 + out[:, :, :ov] / 2
        dm[:, :, prev + 1:start + 384] = out[:, :, max(ov, 0):]
        prev = start + 383
    return dm


MAX_BLEND_WINDOWS = 16     # csrc/window.hip: MAX_STARTS


def _native_plan(model, images, max_batch):
    """The window list [(image index, start column)] of a call that fits ONE forward of the native path -- fp32 device images of the
    model's height whose windows number <= max_batch -- or None (the torch path then runs)."""
    if not images or not hasattr(model, "_engine"):
        return None
    h = images[0].shape[-2]
    plan = [(i, s) for i, im in enumerate(images) for s in window_starts(im.shape[-1])]
    # (countr_window_blend takes at most MAX_BLEND_WINDOWS window positions per image: wider panoramas go to the torch path below)
    if (not plan or len(plan) > min(max_batch, 64) or h != getattr(model, "img_size", 384)
            or any(len(window_starts(im.shape[-1])) > MAX_BLEND_WINDOWS for im in images)
            or any((not im.is_cuda) or im.dtype != torch.float32 or im.dim() != 4 or im.shape[0] != 1 or im.shape[1] != 3 or im.shape[-2] != h
                   or not im.is_contiguous() for im in images)):
        return None
    return plan


def _gather_windows(L, images, plan, h, dst, nb, st):
    """The windows of `plan` cut straight into the batch buffer dst [nb, 3, h, 384] (countr_window_gather); padding rows zeroed."""
    import ctypes as C
    from . import _lib
    nw = len(plan)
    frames = (C.c_void_p * nw)(*[images[i].data_ptr() for i, _s in plan])
    widths = (C.c_int * nw)(*[images[i].shape[-1] for i, _s in plan])
    starts = (C.c_int * nw)(*[s0 for _i, s0 in plan])
    _lib.check(L.countr_window_gather(frames, widths, starts, nw, h, dst.data_ptr(), st), "countr_window_gather")
    if nb > nw:
        dst[nw:].zero_()                       # padding rows only need defined values


@torch.no_grad()
def _native_maps(model, images, boxes, shot_num, max_batch, want_sums, have=None, ahead=None, flags=None):
    """density_maps through the two window kernels of the C ABI (countr_window_gather / countr_window_blend: windows cut straight into
    the engine's input batch, stitched + summed by one launch per image width) when the call fits ONE forward: fp32 device images of
    the model's height whose windows number <= max_batch.  Returns None when it does not apply (the torch path below then runs).
    Pipelined encoder (density_maps_stream): `ahead` = the images of the NEXT call -- if that call takes this path with the same forward
    batch size, their windows are gathered now and their frozen-encoder forward runs beside this call's decoder / density head
    (flags["ahead"] = the engine's ownership token if it did); `have` = the token of the call that ran THIS call's encoder forward."""
    import ctypes as C
    from . import _lib
    plan = _native_plan(model, images, max_batch)
    if plan is None:
        return None
    h = images[0].shape[-2]
    dev = images[0].device
    eng = model._engine()
    L = eng.L
    eng.check_ln_fold(images[plan[0][0]][:, :, :, plan[0][1]:plan[0][1] + 384])     # (first use of a weight set only: may rebuild the plans)
    nb = _bucket(len(plan), max_batch)
    p = eng.plan(nb, shot_num, False)
    img = p.buf["img"]
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    nw = len(plan)
    have = getattr(p, "enc_pipe", None) is not None and eng.pipe_owner(have)      # (`have`: the token of the call that ran our encoder)
    if not have:
        _gather_windows(L, images, plan, h, img, nb, st)
    plan_next = _native_plan(model, ahead, max_batch) if (ahead is not None and getattr(p, "enc_pipe", None) is not None) else None
    if plan_next is not None and (_bucket(len(plan_next), max_batch) != nb or ahead[0].shape[-2] != h or ahead[0].device != dev):
        plan_next = None
    if plan_next is not None:
        _gather_windows(L, ahead, plan_next, h, p.pipe_img[:img.numel()].view(img.shape), nb, st)
    if flags is not None:
        flags["ahead"] = eng.pipe_claim() if plan_next is not None else None
    if shot_num > 0:
        bx = p.buf["boxes"].view(nb, shot_num, 3, 64, 64)
        for j, (i, _s) in enumerate(plan):
            bx[j].copy_(boxes[i][0, :shot_num])
        if nb > nw:
            bx[nw:].zero_()
    if have or plan_next is not None:
        out = eng.forward_loaded_pipelined(nb, shot_num, have, plan_next is not None)
    else:
        out = eng.forward_loaded(nb, shot_num)      # [nb, h, 384], valid until the next forward of this plan
    res, sums = [None] * len(images), [None] * len(images)
    row = 0
    i = 0
    while i < len(images):                      # runs of consecutive images of one width: one blend launch each
        w = images[i].shape[-1]
        j = i
        while j + 1 < len(images) and images[j + 1].shape[-1] == w:
            j += 1
        n = j - i + 1
        stw = window_starts(w)
        if not stw:
            for k in range(i, j + 1):
                res[k] = torch.zeros(h, w, device=dev)          # narrower than a window: the reference's loop never runs
                sums[k] = torch.zeros((), device=dev)
        else:
            dm = torch.empty(n, h, w, device=dev, dtype=torch.float32)
            sm = torch.empty(n, device=dev, dtype=torch.float32) if want_sums else None
            ws = torch.empty(n * int(L.countr_window_blend_blocks(h, w)), device=dev, dtype=torch.float32) if want_sums else None
            arr = (C.c_int * len(stw))(*stw)
            _lib.check(L.countr_window_blend(out[row:].data_ptr(), n, len(stw), arr, h, w, dm.data_ptr(), sm.data_ptr() if want_sums else None,
                                             ws.data_ptr() if want_sums else None, st), "countr_window_blend")
            for k in range(n):
                res[i + k] = dm[k]
                if want_sums:
                    sums[i + k] = sm[k]
            row += n * len(stw)
        i = j + 1
    return (res, sums) if want_sums else res


@torch.no_grad()
def density_maps(model, images, boxes, shot_num, max_batch=32, return_sums=False):
    """Stitched densities of SEVERAL images in as few forwards as possible: images = [[1, 3, 384, w_i], ...], boxes = per image
    [1, >= shot_num, 3, 64, 64] (anything when shot_num == 0) -> [[384, w_i], ...].  Every 384-px window of every image is an
    independent forward of the reference (FSC_test_cross(few-shot).py:326-349, demo_zero.py:49-72), so the windows of all images
    are concatenated and run in chunks of up to max_batch (zero-shot 1920x1080 frames: 8 images x 4 windows = one batch of 32).
    Images of equal width (video frames) are cut into windows and blended together: one strided copy per window position and one
    blend pass for the whole group instead of one per image (the per-image torch launches were ~10 % of the 8-frame call)."""
    shot_num = int(shot_num)
    native = _native_maps(model, images, boxes, shot_num, max_batch, return_sums)
    if native is not None:
        return native
    dev = images[0].device
    h = images[0].shape[-2]
    plan = [(i, s) for i, im in enumerate(images) for s in window_starts(im.shape[-1])]
    # runs of consecutive images with one width: (first image, count, width).  Their windows are consecutive rows of the plan
    # (image-major), so window position k of the whole run is the strided slice rows[k::nwin] -- no index tensors, no host sync
    runs, i = [], 0
    while i < len(images):
        j = i
        while j + 1 < len(images) and images[j + 1].shape[-1] == images[i].shape[-1]:
            j += 1
        runs.append((i, j - i + 1, images[i].shape[-1]))
        i = j + 1
    first_row, r = {}, 0
    for i, im in enumerate(images):
        first_row[i] = r
        r += len(window_starts(im.shape[-1]))
    stacked = {i0: torch.cat(images[i0:i0 + n], 0) for i0, n, w in runs if n > 1 and window_starts(w)}
    outs = []
    for c0 in range(0, len(plan), max_batch):
        chunk = plan[c0:c0 + max_batch]
        nb = _bucket(len(chunk), max_batch)
        wins = (torch.empty if nb == len(chunk) else torch.zeros)(nb, 3, h, 384, device=dev, dtype=torch.float32)   # padding rows only need defined values
        done = set()
        for i0, n, w in runs:                    # whole runs that lie inside this chunk: one strided copy per window position
            st = window_starts(w)
            r0 = first_row[i0] - c0
            if i0 not in stacked or r0 < 0 or r0 + n * len(st) > len(chunk):
                continue
            for k, s0 in enumerate(st):
                wins[r0 + k:r0 + n * len(st):len(st)] = stacked[i0][:, :, :, s0:s0 + 384]
            done.update(range(r0, r0 + n * len(st)))
        for j, (i, s0) in enumerate(chunk):
            if j not in done:
                wins[j] = images[i][0, :, :, s0:s0 + 384]
        if shot_num > 0:
            bx = torch.zeros(nb, shot_num, 3, 64, 64, device=dev, dtype=torch.float32)
            for j, (i, _s) in enumerate(chunk):
                bx[j] = boxes[i][0, :shot_num]
        else:
            bx = torch.zeros(nb, 0, device=dev)
        o = model(wins, bx, shot_num)[:len(chunk)]
        outs.append(o.clone() if c0 + max_batch < len(plan) else o)    # the engine's output buffer lives until the next forward
    outs = torch.cat(outs, 0) if len(outs) > 1 else (outs[0] if outs else None)
    res = [None] * len(images)
    for i0, n, w in runs:
        starts = window_starts(w)
        if not starts:
            for i in range(i0, i0 + n):
                res[i] = torch.zeros(h, w, device=dev)      # narrower than a window: the reference's loop never runs
            continue
        k0 = first_row[i0]
        if n > 1:
            dm = blend_windows_batched(outs[k0:k0 + n * len(starts)].view(n, len(starts), h, 384), starts, w, h)
            for k in range(n):
                res[i0 + k] = dm[k]
        else:
            res[i0] = blend_windows(outs[k0:k0 + len(starts)], starts, w, h)
    return (res, [d.sum() for d in res]) if return_sums else res


@torch.no_grad()
def density_maps_stream(model, groups, shot_num, max_batch=32, return_sums=False):
    """density_maps over a SEQUENCE of calls -- groups = iterable of (images, boxes), e.g. eight video frames each -- with the frozen
    encoder pipelined across them: nothing in a forward depends on another forward (models_mae_cross.py:201-207 under no_grad), so
    while group k's decoder and density head run, group k + 1's windows are already cut and their encoder forward runs on a lane of its
    own (engine.forward_loaded_pipelined); group k + 1 then starts at decoder_embed.  Yields, per group and in order, exactly what
    density_maps returns -- bit-identical maps; 7.70 -> 7.25 ms per 32 windows.  One group of look-ahead: the generator reads group
    k + 1 before it yields group k.  Groups that do not fit the native one-forward path (or whose neighbour needs another forward batch
    size) simply run on their own."""
    shot_num = int(shot_num)
    it = iter(groups)
    cur = next(it, None)
    have = None      # the engine's token for the encoder output computed ahead for `cur` (engine.pipe_claim), or None
    while cur is not None:
        nxt = next(it, None)
        images, boxes = cur
        flags = {}
        res = _native_maps(model, images, boxes, shot_num, max_batch, return_sums, have=have,
                           ahead=(nxt[0] if nxt is not None else None), flags=flags)
        if res is None:
            res = density_maps(model, images, boxes, shot_num, max_batch, return_sums)
        have = flags.get("ahead")
        yield res
        cur = nxt


@torch.no_grad()
def density_map(model, samples, boxes, shot_num, max_batch=32):
    """samples [1, 3, 384, w] -> stitched density [384, w]; all windows run as batched forwards."""
    return density_maps(model, [samples], [boxes], shot_num, max_batch)[0]


def _small_exemplars(pos):
    s_cnt = 0
    for rect in (pos or [])[:3]:
        if rect[2] - rect[0] < 10 and rect[3] - rect[1] < 10:
            s_cnt += 1
    return s_cnt


def _normalise(pred, dm, pos, normalization):
    """Test-time normalisation (FSC_test_cross(few-shot).py:353-359): divide by the mean count inside the exemplar boxes if > 1.8."""
    if normalization and pos:
        e_cnt = sum((dm[r[0]:r[2] + 1, r[1]:r[3] + 1].sum() / 60).item() for r in pos) / 3
        if e_cnt > 1.8:
            pred /= e_cnt
    return pred


@torch.no_grad()
def count_image(model, samples, boxes, shot_num, pos=None, normalization=True, max_s_cnt=1, max_batch=32):
    """Full per-image test path: returns (pred_cnt, density_map).  pos: exemplar rectangles [(y1, x1, y2, x2), ...]."""
    _, _, h, w = samples.shape
    if pos is not None and _small_exemplars(pos) >= max_s_cnt:
        # 3x3 split: each crop is upscaled back to (h, w) and counted on its own (FSC_test_cross(few-shot).py:273-320); the nine
        # upscaled crops are nine independent images for the stitcher, so all their windows share forwards
        crops = []
        for (top, left) in ((0, 0), (h // 3, 0), (0, w // 3), (h // 3, w // 3), (h * 2 // 3, 0), (h * 2 // 3, w // 3),
                            (0, w * 2 // 3), (h // 3, w * 2 // 3), (h * 2 // 3, w * 2 // 3)):
            crop = samples[:, :, top:top + h // 3, left:left + w // 3]
            crops.append(F.interpolate(crop, size=(h, w), mode="bilinear", align_corners=False))
        dms = density_maps(model, crops, [boxes] * 9, shot_num, max_batch)
        pred = sum((d.sum() / 60).item() for d in dms)
        dm = dms[-1]                                   # the reference normalises with the LAST crop's map (its variable is reused)
    else:
        dm = density_map(model, samples, boxes, shot_num, max_batch)
        pred = (dm.sum() / 60).item()
    return _normalise(pred, dm, pos, normalization), dm


@torch.no_grad()
def count_images(model, items, normalization=True, max_s_cnt=1, max_batch=32):
    """Test path over MANY images with windows batched across images: items = [(samples [1,3,384,w], boxes [1,S,3,64,64] or
    empty, pos or None), ...] -> [(pred_cnt, density_map), ...] in input order.  Images are grouped by shot count (one forward
    has one shot_num, models_mae_cross.py:201); images that take the 3x3 split path already fill batches on their own."""
    res = [None] * len(items)
    groups = {}
    for idx, (samples, boxes, pos) in enumerate(items):
        S = boxes.shape[1] if boxes.nelement() > 0 else 0
        if pos is not None and _small_exemplars(pos) >= max_s_cnt:
            res[idx] = count_image(model, samples, boxes, S, pos, normalization, max_s_cnt, max_batch)
        else:
            groups.setdefault(S, []).append(idx)
    for S, idxs in groups.items():
        g0, chunks = 0, []
        while g0 < len(idxs):                           # as many images as fill one forward batch
            g1, nwin = g0, 0
            while g1 < len(idxs) and (g1 == g0 or nwin + len(window_starts(items[idxs[g1]][0].shape[-1])) <= max_batch):
                nwin += len(window_starts(items[idxs[g1]][0].shape[-1]))
                g1 += 1
            chunks.append(idxs[g0:g1])
            g0 = g1
        # consecutive forwards of one shot count: the next chunk's encoder forward runs beside this chunk's decoder / head
        stream = density_maps_stream(model, (([items[i][0] for i in sel], [items[i][1] for i in sel]) for sel in chunks), S, max_batch)
        for sel, dms in zip(chunks, stream):
            for i, dm in zip(sel, dms):
                res[i] = (_normalise((dm.sum() / 60).item(), dm, items[i][2], normalization), dm)
    return res
