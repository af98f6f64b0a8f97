"""HBM-side bytes per kernel CLASS from rocprofv3 --pmc passes over whole train steps (round 5).

    python tools/pmc_class_traffic.py <commit> c3b:<dir with the FETCH_SIZE and WRITE_SIZE passes of `bench.py`> \
                                      c5:<dir ...> > profiles/r05_traffic.json

Every dispatch of a library kernel in the counter_collection.csv files is mapped to the class bench.py times it under
(the vdsprof classes of include/vds.h) and FETCH_SIZE / WRITE_SIZE are averaged over the class's dispatches, so a class
that covers several launch shapes (the GEMM classes: 7-20 shapes per block) gets the per-launch mean of a real step.
bench.py's `roofline.traffic` = (2 * fetch_kb + write_kb) * 1024 of the dominant class: gfx950 FETCH_SIZE reports half
the bytes of a 16-B/lane streaming read (MI355X_MICROARCH.md, HBM)."""
import collections, csv, glob, json, os, re, sys


def klass(name: str):
    n = name
    m = re.search(r"(big::|mid::)?gemm_kernel<(\d+), *(\d+)(?:, *(\d+))?", n)
    if m:
        fmt = int(m.group(4)) if m.group(4) is not None else 0
        if fmt != 0:
            return "gemm_fp8"
        return ("gemm_nt", "gemm_nn", "gemm_tn")[int(m.group(2))]
    if "attn_bwd_dkv16_kernel<96, false>" in n:  # the cross-attention's dK/dV on the 16x16x32 kernel (QPAD = false)
        return "attn_bwd_dkv_plain"
    table = (("attn8_fwd_kernel", "attn_fp8_fwd"), ("attn8_bwd_dkv_kernel", "attn_fp8_dkv"), ("attn8_bwd_dq_kernel", "attn_fp8_dq"),
             ("attn8_delta_kernel", "attn_bwd_delta"), ("attn_fwd16_kernel", "attn_fwd"), ("attn_bwd_dkv16_kernel", "attn_bwd_dkv"),
             ("attn_bwd_dq16_kernel", "attn_bwd_dq"), ("attn_fwd_wide_kernel", "attn_fwd_plain"), ("attn_fwd_kernel", "attn_fwd_plain"),
             ("attn_bwd_dkv_kernel", "attn_bwd_dkv_plain"), ("attn_bwd_dq_kernel", "attn_bwd_dq_plain"),
             ("attn_delta", "attn_bwd_delta"), ("rmsnorm_mod_fwd", "rmsnorm_mod_fwd"), ("rmsnorm_mod_bwd", "rmsnorm_mod_bwd"),
             ("adamw_kernel", "adamw"), ("qkv_rope_fwd", "qkv_rope_fwd"), ("qkv_rope_bwd", "qkv_rope_bwd"),
             ("gate_bwd", "gate_bwd"), ("quant_fp8", "fp8_quant"), ("transpose_fp8", "fp8_quant"), ("absmax", "fp8_quant"))
    for key, c in table:
        if key in n:
            return c
    return None


# Since round 5 the cross-attention's forward and dQ run on the SAME kernels as the self-attention (kv_pad_ones = 2):
# their dispatches differ in what they fetch (512 context rows of K / V instead of 8208), not in name.  The dispatches of
# these symbols are split at the midpoint of a counter's range when it is bimodal (max > 1.4 x min): upper cluster = the
# self-attention class, lower = the cross-attention (`_plain`) class; a counter that does not separate them (both write
# the same O / dQ) counts for both.
SHARED = {"attn_fwd": "attn_fwd_plain", "attn_bwd_dq": "attn_bwd_dq_plain"}


def section(d: str, batch: int):
    acc = collections.defaultdict(lambda: {"FETCH_SIZE": [0.0, 0], "WRITE_SIZE": [0.0, 0], "symbols": collections.Counter()})
    shared = collections.defaultdict(lambda: {"FETCH_SIZE": [], "WRITE_SIZE": [], "sym": None})
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            c = klass(r["Kernel_Name"])
            if c is None or r["Counter_Name"] not in ("FETCH_SIZE", "WRITE_SIZE"):
                continue
            sym = re.sub(r"^void |\(anonymous namespace\)::|\(.*$", "", r["Kernel_Name"])
            if c in SHARED and "16_kernel" in sym:
                shared[c][r["Counter_Name"]].append(float(r["Counter_Value"]))
                shared[c]["sym"] = sym
                continue
            a = acc[c][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
            acc[c]["symbols"][sym] += 1
    for c, v in shared.items():
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            vals = v[ctr]
            if not vals:
                continue
            lo, hi = min(vals), max(vals)
            if hi > 1.4 * lo:
                mid = 0.5 * (lo + hi)
                groups = {c: [x for x in vals if x >= mid], SHARED[c]: [x for x in vals if x < mid]}
            else:
                groups = {c: vals, SHARED[c]: vals if any(x for x in shared[c]["FETCH_SIZE"]) and
                          max(shared[c]["FETCH_SIZE"]) > 1.4 * min(shared[c]["FETCH_SIZE"]) else []}
            for cc, g in groups.items():
                if g:
                    acc[cc][ctr][0] += sum(g)
                    acc[cc][ctr][1] += len(g)
                    acc[cc]["symbols"][v["sym"]] += len(g)
    kernels = {}
    for c, v in sorted(acc.items()):
        if v["FETCH_SIZE"][1] == 0 or v["WRITE_SIZE"][1] == 0:
            continue
        syms = [s for s, _ in v["symbols"].most_common(3)]
        kernels[c] = {"symbol": syms[0] if len(syms) == 1 else " | ".join(syms),
                      "fetch_kb": round(v["FETCH_SIZE"][0] / v["FETCH_SIZE"][1], 1),
                      "write_kb": round(v["WRITE_SIZE"][0] / v["WRITE_SIZE"][1], 1),
                      "launches_averaged": v["FETCH_SIZE"][1]}
    return {"per_gpu_batch": batch, "kernels": kernels}


def csrc_digest():
    """sha256 over the kernel sources the numbers were measured on (bench.py marks the file stale when they differ from the
    tree it runs in: the GPU box has no .git to ask)"""
    import glob
    import hashlib
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "video_diffusion_speedrun_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(root, "*.hip")) + glob.glob(os.path.join(root, "*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


if __name__ == "__main__":
    commit = sys.argv[1]
    out = {"_comment": "HBM-side KiB per launch (FETCH_SIZE, WRITE_SIZE: separate rocprofv3 --pmc passes of `bench.py --steps 1 "
                       "--warmup 1` resp. `--workload c5 --warmup 3`), averaged over every dispatch of a kernel class in the "
                       "profiled run; bytes = (2*fetch_kb + write_kb)*1024 (gfx950 FETCH_SIZE reports half of a 16-B/lane "
                       "streaming read). tools/collect_profiles.sh, tools/pmc_class_traffic.py.",
           "commit": commit, "csrc_sha256": csrc_digest(), "workloads": {}}
    for arg in sys.argv[2:]:
        wl, d = arg.split(":", 1)
        batch = int(os.environ.get("B", 12))
        if d.rsplit(":", 1)[-1].isdigit():  # name:dir:per-GPU batch (the B = 2 section "c3b_b2" of bench.py's small_batch)
            d, b = d.rsplit(":", 1)
            batch = int(b)
        out["workloads"][wl] = section(d, batch)
    print(json.dumps(out, indent=1))
