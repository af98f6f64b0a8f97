#!/usr/bin/env python3
"""First contact with a multi-GPU MI355X node as ONE command (apply_fsdp: model.py:512-542 of the
reference, train.py:323-325, run_debug.sh:12):

    bash tools/first_multigpu.sh                 # all of it: N = 1, 2, 4, 8 (as many as the node has)
    bash tools/first_multigpu.sh --gpus 1,2 --steps 10
    python tools/first_multigpu.py --dry-run     # print the command matrix as JSON lines, run nothing

Runs  bench.py --gpus N  for N in {1,2,4,8} x VDS_COMM_SCHEDULE in {rccl, allpairs} x VDS_AG_PREFETCH in {0, 2}, plus one
run per N in the reshard_after_forward mode (VDS_FSDP_RESHARD=1, window 1)
(N = 1: one plain run + one through the sharding runtime, the knobs do not apply), then  bench.py --comm-only  per N > 1
and schedule, and prints one table: samples/s, scaling efficiency against the N = 1 line, the step time of the slowest
rank, the exposed communication per step (compute-stream stalls on the communication stream, max over ranks), the
communicator's world size and backend.  Every bench line is kept in <out>/runs.jsonl.  Each run is a child process with
its own environment; nothing here touches the GPU itself."""
import argparse
import itertools
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def matrix(gpus, schedules=("rccl", "allpairs"), prefetch=(0, 2), steps=8, warmup=3, workload="c3b", batch=0):
    """the runs, in order: dicts(kind, n, env, argv)"""
    base = ["bench.py", "--steps", str(steps), "--warmup", str(warmup), "--workload", workload,
            "--no-cpu-baseline", "--no-secondary", "--no-small-batch"]
    if batch:
        base += ["--batch", str(batch)]
    runs = []
    for n in gpus:
        if n == 1:
            runs.append({"kind": "step", "n": 1, "env": {}, "argv": base + ["--gpus", "1"]})
            runs.append({"kind": "step", "n": 1, "env": {}, "argv": base + ["--gpus", "1", "--force-shard-runtime"]})
            continue
        for sch, pf in itertools.product(schedules, prefetch):
            runs.append({"kind": "step", "n": n, "env": {"VDS_COMM_SCHEDULE": sch, "VDS_AG_PREFETCH": str(pf)},
                         "argv": base + ["--gpus", str(n)]})
        # the reference's memory-bounded behaviour (reshard_after_forward, model.py:525,541): ring of parameter buffers,
        # depth - 1 more all-gathers per step (fsdp.ReshardRuntime; round 6)
        runs.append({"kind": "step", "n": n, "env": {"VDS_COMM_SCHEDULE": schedules[0], "VDS_AG_PREFETCH": "1",
                                                     "VDS_FSDP_RESHARD": "1"}, "argv": base + ["--gpus", str(n)]})
    for n in gpus:
        if n == 1:
            continue
        for sch in schedules:
            runs.append({"kind": "comm_only", "n": n, "env": {"VDS_COMM_SCHEDULE": sch},
                         "argv": ["bench.py", "--gpus", str(n), "--comm-only", "--steps", "5", "--warmup", "2",
                                  "--workload", workload] + (["--batch", str(batch)] if batch else [])})
    return runs


def last_json(text):
    for line in reversed(text.strip().splitlines()):
        line = line.strip()
        if line.startswith("{") and line.endswith("}"):
            try:
                return json.loads(line)
            except ValueError:
                continue
    return None


def table(results):
    """results: list of (run, bench line | None) -> text"""
    base = next((r["value"] for run, r in results if r and run["kind"] == "step" and run["n"] == 1
                 and "--force-shard-runtime" not in run["argv"]), None)
    rows = [f"{'N':>2} {'schedule':>9} {'prefetch':>8} {'samples/s':>10} {'eff vs N=1':>10} {'ms/step':>9} "
            f"{'exposed comm ms':>15} {'comm world':>10}  backend"]
    for run, r in results:
        if run["kind"] != "step":
            continue
        sch = run["env"].get("VDS_COMM_SCHEDULE", "-")
        pf = run["env"].get("VDS_AG_PREFETCH", "-")
        if "--force-shard-runtime" in run["argv"]:
            sch = "runtime@1"
        if run["env"].get("VDS_FSDP_RESHARD") == "1":
            sch = "reshard"
        if not r:
            rows.append(f"{run['n']:>2} {sch:>9} {pf:>8} {'FAILED':>10}")
            continue
        c = r.get("comm") or {}
        eff = f"{r['value'] / (base * run['n']):.3f}" if base else "-"
        exposed = max(c.get("per_rank_exposed_comm_ms_per_step", [0.0])) if c else 0.0
        rows.append(f"{run['n']:>2} {sch:>9} {pf:>8} {r['value']:>10.3f} {eff:>10} {r['ms_per_step']:>9.1f} "
                    f"{exposed:>15.2f} {str(c.get('communicator_world', '-')):>10}  {c.get('backend', 'single process')}")
    rows.append("")
    rows.append(f"{'N':>2} {'schedule':>9} {'all-gather ms':>14} {'GB/s':>8} {'reduce-scatter ms':>18} {'GB/s':>8}   (collectives of one step, alone)")
    for run, r in results:
        if run["kind"] != "comm_only":
            continue
        if not r:
            rows.append(f"{run['n']:>2} {run['env']['VDS_COMM_SCHEDULE']:>9} FAILED")
            continue
        t = r["totals"]
        rows.append(f"{run['n']:>2} {run['env']['VDS_COMM_SCHEDULE']:>9} {t['all_gather']['ms']:>14.2f} "
                    f"{t['all_gather']['GB/s'] or 0:>8.1f} {t['reduce_scatter']['ms']:>18.2f} {t['reduce_scatter']['GB/s'] or 0:>8.1f}")
    return "\n".join(rows)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", default="1,2,4,8")
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c3b")
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--out", default=os.path.join(REPO, "gpurun_out", "first_multigpu"))
    ap.add_argument("--dry-run", action="store_true")
    ap.add_argument("--timeout", type=int, default=900, help="seconds per run")
    args = ap.parse_args()
    gpus = [int(x) for x in args.gpus.split(",") if x]
    runs = matrix(gpus, steps=args.steps, warmup=args.warmup, workload=args.workload, batch=args.batch)
    if args.dry_run:
        for r in runs:
            print(json.dumps(r))
        return 0
    os.makedirs(args.out, exist_ok=True)
    results = []
    with open(os.path.join(args.out, "runs.jsonl"), "w") as log:
        for r in runs:
            env = dict(os.environ, **r["env"])
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            print(f"[first_multigpu] N={r['n']} {r['kind']} {r['env']} ...", file=sys.stderr, flush=True)
            try:
                p = subprocess.run([sys.executable] + r["argv"], cwd=REPO, env=env, capture_output=True, text=True,
                                   timeout=args.timeout)
                line = last_json(p.stdout) if p.returncode == 0 else None
                err = None if line else (p.stderr or p.stdout)[-800:]
            except subprocess.TimeoutExpired:
                line, err = None, f"timeout after {args.timeout} s"
            results.append((r, line))
            log.write(json.dumps({"run": r, "line": line, "error": err}) + "\n")
            log.flush()
    text = table(results)
    with open(os.path.join(args.out, "table.txt"), "w") as f:
        f.write(text + "\n")
    print(text)
    return 0 if all(l for _, l in results) else 1


if __name__ == "__main__":
    sys.exit(main())
