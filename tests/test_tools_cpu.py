"""Host-side checks of the measurement tooling that needs no GPU: the command matrix of tools/first_multigpu.py (the
one-command first run on a multi-GPU node) against bench.py's own argument parser, its table, and the per-class traffic
summariser of the rocprofv3 PMC passes."""
import importlib.util
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REPO, path))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_first_multigpu_matrix_is_what_bench_py_accepts():
    fm = _load("first_multigpu", "tools/first_multigpu.py")
    sys.path.insert(0, REPO)
    import bench
    runs = fm.matrix([1, 2, 4, 8])
    steps = [r for r in runs if r["kind"] == "step"]
    comm = [r for r in runs if r["kind"] == "comm_only"]
    # N = 1: plain + through the sharding runtime; N > 1: 2 schedules x 2 all-gather windows + the reshard_after_forward
    # mode; comm-only: N > 1 x 2 schedules
    assert len(steps) == 2 + 3 * 5 and len(comm) == 3 * 2
    plain = [r for r in steps if r["n"] > 1 and "VDS_FSDP_RESHARD" not in r["env"]]
    assert {(r["n"], r["env"].get("VDS_COMM_SCHEDULE"), r["env"].get("VDS_AG_PREFETCH")) for r in plain} == \
        {(n, s, p) for n in (2, 4, 8) for s in ("rccl", "allpairs") for p in ("0", "2")}
    assert sorted(r["n"] for r in steps if r["env"].get("VDS_FSDP_RESHARD") == "1") == [2, 4, 8]
    parser = bench.build_parser()
    for r in runs:
        assert r["argv"][0] == "bench.py"
        a = parser.parse_args(r["argv"][1:])  # raises SystemExit on a flag bench.py does not know
        assert a.gpus == r["n"] and a.workload == "c3b"
        assert a.comm_only == (r["kind"] == "comm_only")
        assert set(r["env"]) <= {"VDS_COMM_SCHEDULE", "VDS_AG_PREFETCH", "VDS_FSDP_RESHARD"}
        if r["kind"] == "step":  # the scaling points must be the headline measurement alone
            assert a.no_cpu_baseline and a.no_secondary and a.no_small_batch
    assert sum("--force-shard-runtime" in r["argv"] for r in runs) == 1
    # a node with fewer GPUs
    assert {r["n"] for r in fm.matrix([1, 2])} == {1, 2}


def test_first_multigpu_table_reports_efficiency_and_exposed_comm():
    fm = _load("first_multigpu", "tools/first_multigpu.py")
    runs = fm.matrix([1, 2])
    lines = []
    for r in runs:
        if r["kind"] == "comm_only":
            lines.append({"totals": {"all_gather": {"ms": 10.0, "GB/s": 227.0}, "reduce_scatter": {"ms": 20.0, "GB/s": 226.5}}})
        elif r["n"] == 1:
            lines.append({"value": 14.0, "ms_per_step": 857.0})
        else:
            lines.append({"value": 26.6, "ms_per_step": 902.0,
                          "comm": {"per_rank_exposed_comm_ms_per_step": [3.0, 4.5], "communicator_world": 2,
                                   "backend": "vds_comm (RCCL from csrc/comm.hip)"}})
    lines[3] = None  # one failed run must not hide the others
    text = fm.table(list(zip(runs, lines)))
    assert "0.950" in text and "4.50" in text and "FAILED" in text and "vds_comm" in text
    assert fm.last_json("noise\n{\"a\": 1}\ntrailing") == {"a": 1}


def test_pmc_class_traffic_maps_kernels_to_bench_classes(tmp_path):
    pt = _load("pmc_class_traffic", "tools/pmc_class_traffic.py")
    k = pt.klass
    assert k("void (anonymous namespace)::big::gemm_kernel<0, 2, 0, 256, 2>((anonymous namespace)::GemmP)") == "gemm_nt"
    assert k("void (anonymous namespace)::big::gemm_kernel<1, 3, 0, 256, 2>(GemmP)") == "gemm_nn"
    assert k("void (anonymous namespace)::mid::gemm_kernel<2, 4>(GemmP)") == "gemm_tn"
    assert k("void (anonymous namespace)::big::gemm_kernel<0, 1, 1, 256, 2>(GemmP)") == "gemm_fp8"
    assert k("void (anonymous namespace)::big::gemm_kernel<2, 4, 3, 256, 2>(GemmP)") == "gemm_fp8"
    assert k("void (anonymous namespace)::attn_bwd_dkv16_kernel<96>(AttnP)") == "attn_bwd_dkv"
    assert k("void (anonymous namespace)::attn_bwd_dkv_kernel<96, 80, false>(AttnP)") == "attn_bwd_dkv_plain"
    assert k("void (anonymous namespace)::attn8_bwd_dq_kernel<72, 4>(Attn8P)") == "attn_fp8_dq"
    assert k("void at::native::vectorized_elementwise_kernel<4, FillFunctor<float>>(int)") is None
    sys.path.insert(0, REPO)
    import bench
    assert {c for c in (k(n) for n in ("adamw_kernel(", "gate_bwd_kernel<3, -1>(", "rmsnorm_mod_bwd_kernel<3, false>(",
                                        "qkv_rope_bwd_tok_kernel<3, -1>(", "attn_delta_tokmajor_kernel("))} <= set(bench.PROF_NAMES)
    # two passes over two dispatches of one class -> per-launch means, in the layout bench.roofline_of reads
    for c, vals in (("FETCH_SIZE", (100.0, 300.0)), ("WRITE_SIZE", (50.0, 70.0))):
        d = tmp_path / "c3b" / c
        d.mkdir(parents=True)
        with open(d / "x_counter_collection.csv", "w") as f:
            f.write("Kernel_Name,Counter_Name,Counter_Value\n")
            for v in vals:
                f.write(f"\"void (anonymous namespace)::mid::gemm_kernel<2, 4>((anonymous namespace)::GemmP)\",{c},{v}\n")
    sec = pt.section(str(tmp_path / "c3b"), 12)
    assert sec["per_gpu_batch"] == 12 and sec["kernels"]["gemm_tn"]["fetch_kb"] == 200.0
    assert sec["kernels"]["gemm_tn"]["write_kb"] == 60.0 and sec["kernels"]["gemm_tn"]["launches_averaged"] == 2
