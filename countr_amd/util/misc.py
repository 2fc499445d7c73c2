"""Minimal harness helpers with the reference's behaviour (util/misc.py:225-257, 304-328, 363-421, 424-432)."""
import os

import torch
import torch.distributed as dist


def is_dist():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if is_dist() else 1


def get_rank():
    return dist.get_rank() if is_dist() else 0


def is_main_process():
    return get_rank() == 0


def init_distributed_mode(args):
    """The reference's three launcher conventions (util/misc.py:225-257), backend nccl == RCCL:
    --dist_on_itp: OpenMPI (OMPI_COMM_WORLD_RANK / _SIZE / _LOCAL_RANK, tcp://MASTER_ADDR:MASTER_PORT; the variables torchrun would have
    set are exported like the reference does); RANK / WORLD_SIZE / LOCAL_RANK (torch.distributed.run); SLURM_PROCID (gpu = rank modulo
    the visible devices; the world size comes from --world_size, as in the reference).  Anything else: single process."""
    env = os.environ
    if getattr(args, "dist_on_itp", False):
        args.rank = int(env["OMPI_COMM_WORLD_RANK"])
        args.world_size = int(env["OMPI_COMM_WORLD_SIZE"])
        args.gpu = int(env["OMPI_COMM_WORLD_LOCAL_RANK"])
        args.dist_url = "tcp://%s:%s" % (env["MASTER_ADDR"], env["MASTER_PORT"])
        env["LOCAL_RANK"], env["RANK"], env["WORLD_SIZE"] = str(args.gpu), str(args.rank), str(args.world_size)
    elif "RANK" in env and "WORLD_SIZE" in env:
        args.rank = int(env["RANK"])
        args.world_size = int(env["WORLD_SIZE"])
        args.gpu = int(env.get("LOCAL_RANK", 0))
    elif "SLURM_PROCID" in env:
        args.rank = int(env["SLURM_PROCID"])
        args.world_size = int(getattr(args, "world_size", 1) or 1)
        args.gpu = args.rank % max(torch.cuda.device_count(), 1)
    else:
        args.distributed = False
        args.rank, args.world_size, args.gpu = 0, 1, 0
        return
    if args.world_size <= 1:          # (a one-rank "job": nothing to exchange, no process group)
        args.distributed = False
        args.rank, args.world_size = 0, 1
        torch.cuda.set_device(args.gpu % max(torch.cuda.device_count(), 1))
        return
    args.distributed = True
    # COUNTR_DIST_BACKEND=gloo: dry run of an N-rank job on fewer GPUs (RCCL refuses two ranks per device); ranks then share devices
    backend = env.get("COUNTR_DIST_BACKEND", "nccl")
    if backend != "nccl":
        args.gpu %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(args.gpu)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    print("| distributed init (rank %d): %s, gpu %d" % (args.rank, getattr(args, "dist_url", "env://"), args.gpu), flush=True)
    dist.init_process_group(backend=backend, init_method=getattr(args, "dist_url", "env://"), world_size=args.world_size, rank=args.rank)
    dist.barrier()


def all_reduce_mean(x):
    if get_world_size() > 1:
        t = torch.tensor(float(x), device="cuda")
        dist.all_reduce(t)
        return (t / get_world_size()).item()
    return float(x)


def save_model(args, epoch, model_without_ddp, optimizer_state, suffix="", scaler_state=None):
    """Checkpoint dict {'model','optimizer','epoch','scaler','args'} named checkpoint__<suffix>.pth (util/misc.py:304-328).
    scaler_state: the step's GradScaler state_dict (fp16 mode: trainer.scaler_state()).  bf16 / fp32 have no scaler and the key is
    OMITTED: the reference's resume reads it only `if 'scaler' in checkpoint` (util/misc.py:418-419) and its GradScaler.load_state_dict
    raises on an empty dict, so an empty entry would make the file unloadable there."""
    if not is_main_process() or not args.output_dir:
        return None
    os.makedirs(args.output_dir, exist_ok=True)
    path = os.path.join(args.output_dir, "checkpoint%s.pth" % ("__" + suffix if suffix else ""))
    ckpt = {"model": {k: v.detach().cpu() for k, v in model_without_ddp.state_dict().items()},
            "optimizer": optimizer_state, "epoch": epoch, "args": vars(args)}
    if scaler_state:
        ckpt["scaler"] = dict(scaler_state)
    torch.save(ckpt, path)
    return path


def _checkpoint_path(args):
    """The reference hands args.resume straight to torch.load (util/misc.py:338-376): a missing file raises.  Here an EMPTY
    --resume means "start from the model's own initialisation"; a non-empty path that does not exist is an error -- with the
    encoder frozen, silently finetuning a randomly initialised encoder would still write plausible-looking checkpoints."""
    if not args.resume:
        return None
    if not os.path.exists(args.resume):
        raise FileNotFoundError("--resume %r does not exist (pass --resume '' to start from the model's initialisation)" % args.resume)
    return args.resume


def load_model_FSC(args, model_without_ddp):
    """util/misc.py:363-376: strict=False, pos_embed dropped on shape mismatch."""
    if _checkpoint_path(args) is None:
        return None
    ckpt = torch.load(args.resume, map_location="cpu", weights_only=False)
    sd = ckpt["model"] if "model" in ckpt else ckpt
    if "pos_embed" in sd and sd["pos_embed"].shape != model_without_ddp.state_dict()["pos_embed"].shape:
        print("Removing key pos_embed from pretrained checkpoint")
        del sd["pos_embed"]
    model_without_ddp.load_state_dict(sd, strict=False)
    print("Resume checkpoint %s" % args.resume)
    return ckpt


def load_model(args, model_without_ddp):
    """util/misc.py:338-361 (pretraining resume, starts from MAE ImageNet weights FSC_pretrain.py:80): strict=False, both
    pos-embeds dropped on shape mismatch; returns the checkpoint (the caller restores the flat AdamW state / epoch)."""
    if _checkpoint_path(args) is None:
        return None
    ckpt = torch.load(args.resume, map_location="cpu", weights_only=False)
    sd = ckpt["model"]
    have = model_without_ddp.state_dict()
    for k in ("pos_embed", "decoder_pos_embed"):
        if k in sd and sd[k].shape != have[k].shape:
            print("Removing key %s from pretrained checkpoint" % k)
            del sd[k]
    model_without_ddp.load_state_dict(sd, strict=False)
    print("Resume checkpoint %s" % args.resume)
    return ckpt
