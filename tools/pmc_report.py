"""Per-(kernel, grid) and per-family summary of the rocprofv3 --pmc passes of tools/pmc_step.sh.
usage: pmc_report.py <out prefix> <db> [<db> ...]   ->  <prefix>_pmc_groups.json + a text table on stdout.

Families as in bench.py::family_breakdown: linear (nn.Linear fwd / dgrad / wgrad GEMMs), conv (3x3 implicit-GEMM convolutions),
attention, other.  HBM-side traffic per launch = FETCH_SIZE x 2 + WRITE_SIZE (KB): gfx950's rocprofv3 tallies the 128-byte requests of
wide coalesced reads at 64 B (MI355X_MICROARCH.md, section HBM); WRITE_SIZE as reported.  Matrix-pipe share = SQ_VALU_MFMA_BUSY_CYCLES /
SQ_BUSY_CYCLES / 32 (the first sums over 1024 SIMDs, the second over 32 shader engines)."""
import collections, json, re, sqlite3, sys


def family(name):
    if "fa_fwd" in name or "flash_attn" in name:
        return "attention"
    m = re.match(r".*g256_kernel<(true|false)", name)
    if m:
        return "conv" if m.group(1) == "true" else "linear"
    if "cwg_group_kernel" in name:      # a block's nn.Linear weight gradients in one launch
        return "linear"
    m = re.match(r".*cwg_kernel<\d+, (true|false)", name)
    if m:
        return "linear" if m.group(1) == "true" else "conv"
    m = re.match(r".*lin_kernel<\d+, \d+, \d+, \d+, (true|false)", name)
    if m:
        return "conv" if m.group(1) == "true" else "linear"
    m = re.match(r".*gemm_kernel<[^,]+, (\d+), (\d+)", name)
    if m:
        return "conv" if (m.group(1) == "2" or m.group(2) == "3") else "linear"
    if "conv3x3_c3" in name:
        return "conv"
    return "other"


def load(db):
    cur = sqlite3.connect(db).cursor()
    q = """select k.kernel_name, d.grid_size_x, d.workgroup_size_x, c.name, avg(p.value), count(*) from rocpd_pmc_event p
     join rocpd_info_pmc c on p.pmc_id = c.id join rocpd_kernel_dispatch d on p.event_id = d.event_id
     join rocpd_info_kernel_symbol k on d.kernel_id = k.id group by k.kernel_name, d.grid_size_x, c.name"""
    rows = list(cur.execute(q))
    dur = {}
    try:
        for kn, gx, avg_ns, n in cur.execute("""select k.kernel_name, d.grid_size_x, avg(d.end - d.start), count(*) from rocpd_kernel_dispatch d
                join rocpd_info_kernel_symbol k on d.kernel_id = k.id group by k.kernel_name, d.grid_size_x"""):
            dur[(kn, gx)] = (avg_ns, n)
    except Exception:  # noqa: BLE001
        pass
    return rows, dur


def demangle(names):
    import shutil
    import subprocess
    tool = shutil.which("c++filt") or shutil.which("llvm-cxxfilt") or "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"
    try:
        out = subprocess.run([tool], input="\n".join(re.sub(r"\.kd$", "", n) for n in names), capture_output=True,
                             text=True, timeout=60).stdout.splitlines()
        if len(out) == len(names):
            return dict(zip(names, out))
    except Exception:  # noqa: BLE001
        pass
    return {n: n for n in names}


def main():
    if sys.argv[1] == "--json":      # re-run the classification / table on a saved groups file (e.g. on a host that has a demangler)
        saved = json.load(open(sys.argv[2]))
        groups = collections.defaultdict(dict)
        for d in saved["groups"]:
            g = {k: v for k, v in d.items() if k.isupper() or k in ("n", "wg", "us_profiled")}
            groups[(d.get("mangled", d["kernel"]), d["grid_threads"])] = g
        return report(sys.argv[3], groups)
    prefix, dbs = sys.argv[1], sys.argv[2:]
    groups = collections.defaultdict(dict)
    for db in dbs:
        rows, dur = load(db)
        for kn, gx, wx, cn, v, n in rows:
            g = groups[(kn, gx)]
            g[cn] = v
            g["n"] = n
            g["wg"] = wx
        for (kn, gx), (avg_ns, n) in dur.items():
            if (kn, gx) in groups:
                groups[(kn, gx)].setdefault("us_profiled", avg_ns / 1e3)
    return report(prefix, groups)


def report(prefix, groups):
    nsteps = None
    dm = demangle(sorted({k[0] for k in groups}))
    out = []
    for (kn, gx), g in groups.items():
        name = re.sub(r"^void ", "", dm[kn]).replace("(anonymous namespace)::", "")
        name = re.sub(r"\(.*$", "", name)
        d = dict(g)
        d.update(kernel=name, mangled=kn, grid_threads=gx, wgs=(gx // g["wg"]) if g.get("wg") else None, family=family(name))
        if "FETCH_SIZE" in d or "WRITE_SIZE" in d:
            d["traffic_bytes"] = (2.0 * d.get("FETCH_SIZE", 0.0) + d.get("WRITE_SIZE", 0.0)) * 1024.0
        if d.get("SQ_BUSY_CYCLES"):
            # SQ_VALU_MFMA_BUSY_CYCLES sums over the 1024 SIMDs, SQ_BUSY_CYCLES over the 32 shader engines (8 XCDs x 4): per SIMD and per
            # busy cycle the ratio is / 32 (profiles/r2_attention_pmc.txt: 7776 / 32455 per SIMD)
            d["mfma_busy_share"] = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / d["SQ_BUSY_CYCLES"] / 32.0
        if d.get("SQ_LDS_IDX_ACTIVE"):
            d["lds_conflict_share"] = d.get("SQ_LDS_BANK_CONFLICT", 0.0) / d["SQ_LDS_IDX_ACTIVE"]
        if d.get("SQ_WAVE_CYCLES"):
            d["wait_share"] = d.get("SQ_WAIT_ANY", 0.0) / d["SQ_WAVE_CYCLES"]
            d["issue_stall_share"] = d.get("SQ_WAIT_INST_ANY", 0.0) / d["SQ_WAVE_CYCLES"]
        if "adamw_kernel" in name:
            nsteps = d["n"]
        out.append(d)
    nsteps = nsteps or 1
    fam = collections.defaultdict(lambda: {"traffic_bytes_per_step": 0.0, "launches_per_step": 0.0})
    for d in out:
        f = fam[d["family"]]
        f["traffic_bytes_per_step"] += d.get("traffic_bytes", 0.0) * d["n"] / nsteps
        f["launches_per_step"] += d["n"] / nsteps
    json.dump({"steps_profiled": nsteps, "families": fam, "groups": out,
               "correction": "traffic = FETCH_SIZE x 2 + WRITE_SIZE (KB -> bytes); FETCH_SIZE / WRITE_SIZE from separate --pmc passes"},
              open(prefix + "_pmc_groups.json", "w"), indent=1)
    print("steps profiled: %d" % nsteps)
    for k, f in sorted(fam.items()):
        print("family %-10s %7.1f launches/step   traffic %9.1f MB/step" % (k, f["launches_per_step"], f["traffic_bytes_per_step"] / 1e6))
    print()
    print("%-66s %6s %5s %8s %9s %9s %9s %6s %6s %6s %6s" % ("kernel", "wgs", "n/st", "us", "fetchx2MB", "writeMB", "trafficMB", "mfma%", "wait%", "stall%", "ldscf%"))
    for d in sorted(out, key=lambda d: -(d.get("us_profiled", 0) * d["n"])):
        if d["family"] == "other" and d.get("us_profiled", 0) * d["n"] / nsteps < 20:
            continue
        print("%-66s %6s %5.1f %8.1f %9.2f %9.2f %9.2f %6.1f %6.1f %6.1f %6.1f" % (
            d["kernel"][:66], d["wgs"], d["n"] / nsteps, d.get("us_profiled", 0.0), 2 * d.get("FETCH_SIZE", 0) * 1024 / 1e6, d.get("WRITE_SIZE", 0) * 1024 / 1e6,
            d.get("traffic_bytes", 0) / 1e6, 100 * d.get("mfma_busy_share", 0), 100 * d.get("wait_share", 0), 100 * d.get("issue_stall_share", 0),
            100 * d.get("lds_conflict_share", 0)))


if __name__ == "__main__":
    main()
