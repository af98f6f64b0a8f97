"""summarise rocprofv3 --pmc counter_collection.csv files per kernel (last dispatch of each kernel)"""
import csv, collections, re, sys
res = collections.defaultdict(dict)
for f in sys.argv[1:]:
    for r in csv.DictReader(open(f)):
        m = re.search(r"(attn8?_\w+|(?:big::|mid::)?gemm_kernel<[\d, ]+>|\w+_kernel)", r["Kernel_Name"])
        if not m:
            continue
        k = m.group(1)
        res[k][r["Counter_Name"]] = float(r["Counter_Value"])
        res[k]["dur_us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        res[k]["vgpr"] = r["VGPR_Count"]
        res[k]["lds"] = r["LDS_Block_Size"]
for k, v in res.items():
    print(k, " ".join(f"{a}={b:.4g}" if isinstance(b, float) else f"{a}={b}" for a, b in sorted(v.items())))
    wc = v.get("SQ_WAVE_CYCLES", 0)
    g = lambda n: v.get(n, float("nan"))
    if wc:
        print("   per wave-cycle: wait_any %.2f wait_inst_any %.2f active_any %.2f active_valu %.2f active_lds %.2f wait_inst_lds %.2f"
              % (g("SQ_WAIT_ANY") / wc, g("SQ_WAIT_INST_ANY") / wc, g("SQ_ACTIVE_INST_ANY") / wc, g("SQ_ACTIVE_INST_VALU") / wc,
                 g("SQ_ACTIVE_INST_LDS") / wc, g("SQ_WAIT_INST_LDS") / wc))
    if v.get("SQ_INSTS_MFMA"):
        print("   valu/mfma %.1f  lds_inst/mfma %.2f  bank_conflict/lds_idx_active %.2f  mfma_busy_us_per_simd@2.1GHz %.0f (dur %.0f us)"
              % (g("SQ_INSTS_VALU") / g("SQ_INSTS_MFMA"), g("SQ_INSTS_LDS") / g("SQ_INSTS_MFMA"),
                 g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE"), g("SQ_VALU_MFMA_BUSY_CYCLES") / 1024 / 2100, g("dur_us")))
    if v.get("GRBM_GUI_ACTIVE"):
        print("   clock ~ %.2f GHz" % (g("GRBM_GUI_ACTIVE") / 8 / (g("dur_us") * 1e3)))
