"""Turns the counter groups of tools/pmc_step.sh (gpurun_out/<tag>_pmc_groups.json) into the round's evidence files:
profiles/<tag>_family_traffic.json (read by bench.py into roofline_families.{linear,conv}.traffic), <tag>_linear_pmc.txt, <tag>_conv_pmc.txt,
<tag>_attention_pmc.txt, <tag>_attention_traffic.json.  usage: python tools/pmc_postprocess.py [tag] [algorithmic bytes json from the bench line]

traffic = FETCH_SIZE x 2 + WRITE_SIZE (KB): gfx950's rocprofv3 tallies the 128-byte requests of wide coalesced reads at 64 B
(MI355X_MICROARCH.md, section HBM); the two counters come from separate passes.  They count requests on the L2's memory side, i.e.
Infinity-Cache hits are included: "traffic" is fabric traffic, an upper bound of HBM traffic."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r4"
src = os.path.join(ROOT, "gpurun_out", tag + "_pmc_groups.json")
tmp = "/tmp/%s_re" % tag
subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_report.py"), "--json", src, tmp], stdout=subprocess.DEVNULL)
d = json.load(open(tmp + "_pmc_groups.json"))
groups, nsteps = d["groups"], d["steps_profiled"]
MB = 1e6
B, rows = 8, 8 * 576

# algorithmic bytes of the launches that own the step (B = 8): operands once + results once
ALG = {
    ("lin_kernel<1, 4, 0, 2, false, 2, true, 3>", 256): ("encoder qkv 4608x2304x768 (192x256 tiles, LN consumer)", rows * 768 * 2 + 2304 * 768 * 2 + rows * 2304 * 2 + rows * 12 * 8),
    ("lin_kernel<1, 4, 2, 3, false, 1, true, 2>", 256): ("encoder proj / fc2 (average of the two: LN producers, fp32 residual in + out, bf16 copy)",
                                                        ((rows * 768 * 2 + 768 * 768 * 2) + (rows * 3072 * 2 + 768 * 3072 * 2)) / 2 + rows * 768 * (4 + 4 + 2) + rows * 12 * 8),
    ("g256_kernel<false, 1, true>", 256): ("encoder fc1 4608x3072x768 (256x256 tiles, LN consumer, GELU)", rows * 768 * 2 + 3072 * 768 * 2 + rows * 3072 * 2 + rows * 12 * 8),
    ("g256_kernel<true, 0, false>", 1024): ("3x3 convolution 192x192, 256 -> 256, forward / dgrad: the 8/9 of the rows on 256x256 tiles",
                                            (B * 192 * 192 * 256 * 2 * 2) * 8 / 9 + 256 * 2304 * 2),
    ("lin_kernel<1, 4, 0, 3, true, 2, false, 2>", 256): ("... its last 1/9 of the rows on 128x256 tiles (split rounds)", (B * 192 * 192 * 256 * 2 * 2) / 9 + 256 * 2304 * 2),
    ("lin_kernel<1, 4, 0, 3, true, 2, false, 2>", 576): ("3x3 convolution 96x96, 256 -> 256, forward / dgrad (128x256 tiles)", B * 96 * 96 * 256 * 2 * 2 + 256 * 2304 * 2),
    ("lin_kernel<1, 4, 0, 3, true, 2, false, 2>", 144): ("3x3 convolution 48x48, 256 -> 256, forward / dgrad", B * 48 * 48 * 256 * 2 * 2 + 256 * 2304 * 2),
    ("cwg_kernel<2, false>", 252): ("3x3 convolution weight gradients (average of 192x192 / 96x96 / 48x48 / 24x24: both maps + fp32 dW once)",
                                    (B * (192 * 192 + 96 * 96 + 48 * 48) * 256 * 4 + B * 24 * 24 * (512 + 256) * 2 + 3 * 256 * 2304 * 4 + 256 * 4608 * 4) / 4),
    ("fa_fwd_pipe_kernel<64, false, 0, true>", 480): ("encoder attention core (q, k, v in, o out)", 28311552),
}


def table(fam, path, title):
    sel = sorted([g for g in groups if g["family"] == fam], key=lambda g: -(g.get("us_profiled", 0) * g["n"]))
    tot = sum(g.get("traffic_bytes", 0) * g["n"] for g in sel) / nsteps
    with open(path, "w") as f:
        f.write(title + "\n")
        f.write("source: bash tools/pmc_step.sh %s  (rocprofv3 --kernel-trace --pmc <group>, four separate passes over `python bench.py --steps 3 "
                "--warmup 2 --reps 1 --no-graph --plain`: every launch belongs to a headline step; %d steps profiled), summarised by "
                "tools/pmc_report.py + tools/pmc_postprocess.py\n" % (tag, nsteps))
        f.write("traffic = FETCH_SIZE x 2 + WRITE_SIZE (fabric side of the L2: Infinity-Cache hits included); mfma% = SQ_VALU_MFMA_BUSY_CYCLES / "
                "SQ_BUSY_CYCLES / 32; wait% / stall% = SQ_WAIT_ANY / SQ_WAIT_INST_ANY over SQ_WAVE_CYCLES; ldscf% = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE\n")
        f.write("family total: %.1f MB of traffic per step in %.1f launches\n\n" % (tot / MB, sum(g["n"] for g in sel) / nsteps))
        f.write("%-58s %5s %5s %8s %9s %8s %9s %9s %6s %6s %6s %6s %6s\n" % ("kernel", "wgs", "n/st", "us", "fetchx2MB", "writeMB", "trafficMB", "algMB", "t/alg", "mfma%", "wait%", "stall%", "ldscf%"))
        notes = []
        for g in sel:
            key = (g["kernel"], g["wgs"])
            alg = ALG.get(key)
            t = g.get("traffic_bytes", 0)
            f.write("%-58s %5s %5.1f %8.1f %9.2f %8.2f %9.2f %9s %6s %6.1f %6.1f %6.1f %6.1f\n" % (
                g["kernel"][:58], g["wgs"], g["n"] / nsteps, g.get("us_profiled", 0), 2 * g.get("FETCH_SIZE", 0) * 1024 / MB, g.get("WRITE_SIZE", 0) * 1024 / MB,
                t / MB, ("%.2f" % (alg[1] / MB)) if alg else "-", ("%.2f" % (t / alg[1])) if alg else "-", 100 * g.get("mfma_busy_share", 0),
                100 * g.get("wait_share", 0), 100 * g.get("issue_stall_share", 0), 100 * g.get("lds_conflict_share", 0)))
            if alg:
                notes.append("  %s wgs=%s: %s" % (g["kernel"], g["wgs"], alg[0]))
        f.write("\nshapes:\n" + "\n".join(notes) + "\n")
    return tot, sum(g["n"] for g in sel) / nsteps


os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
P = lambda n: os.path.join(ROOT, "profiles", "%s_%s" % (tag, n))
lin = table("linear", P("linear_pmc.txt"), "nn.Linear family (forward, input and weight gradients) inside the finetune step, B = 8, bf16")
conv = table("conv", P("conv_pmc.txt"), "3x3 convolution family (forward, dgrad, wgrad; + the 3 -> 64 direct convolution) inside the finetune step, B = 8, bf16")
att = table("attention", P("attention_pmc.txt"), "fused self-attention (encoder forward dh = 64; decoder forward + backward dh = 32) inside the finetune step, B = 8, bf16")
alg = {}
if len(sys.argv) > 2:
    line = json.load(open(sys.argv[2]))
    for k in ("linear", "conv"):
        alg[k] = line["roofline_families"][k].get("algorithmic_bytes")
fam = {"source": "tools/pmc_step.sh %s + tools/pmc_postprocess.py: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over the eager "
                 "finetune step (bench.py --plain --no-graph), all launches of the family per step; traffic = FETCH_SIZE x 2 + WRITE_SIZE" % tag,
       "steps_profiled": nsteps,
       "collected_on_head": subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], capture_output=True, text=True).stdout.strip() or None,
       "tree_dirty": bool(subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--", "countr_amd", "bench.py"], capture_output=True, text=True).stdout.strip())}
for k, (tot, nl) in (("linear", lin), ("conv", conv), ("attention", att)):
    if not isinstance(fam.get(k, {}), dict):
        continue
    fam[k] = {"traffic_bytes_per_step": tot, "launches_per_step": nl}
    if alg.get(k):
        fam[k]["algorithmic_bytes_per_step"] = alg[k]
        fam[k]["traffic_over_algorithmic"] = tot / alg[k]
json.dump(fam, open(P("family_traffic.json"), "w"), indent=1)
fa = [g for g in groups if g["kernel"].startswith("fa_fwd_pipe_kernel<64") and g["wgs"] == 480]
if fa:
    g = fa[0]
    json.dump({"kernel": g["kernel"] + " (K/V staged by MUBUF LDS-DMA)", "batch": 8, "FETCH_SIZE_KB_mean": g.get("FETCH_SIZE"), "WRITE_SIZE_KB_mean": g.get("WRITE_SIZE"),
               "correction": "FETCH_SIZE x2 (gfx950 tallies 128-byte requests of wide coalesced reads at 64 B: MI355X_MICROARCH.md section HBM); WRITE_SIZE as reported",
               "bytes_per_launch": int(g.get("traffic_bytes", 0)), "algorithmic_bytes": 28311552, "launches": g["n"],
               "mfma_busy_share": g.get("mfma_busy_share"), "wait_share": g.get("wait_share"), "issue_stall_share": g.get("issue_stall_share"),
               "command": "bash tools/pmc_step.sh %s (separate --pmc passes, --kernel-trace only)" % tag}, open(P("attention_traffic.json"), "w"), indent=1)
print(json.dumps(fam, indent=1))
