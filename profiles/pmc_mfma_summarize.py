"""MFMA utilisation of the bf16-A GEMM launches of profiles/pmc_gemm_shapes.py from a rocprofv3 PMC pass:

    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d gpurun_out/pmc_mfma --output-format csv -- python3 profiles/pmc_gemm_shapes.py
    python profiles/pmc_mfma_summarize.py <counter_collection.csv> <kernel_trace.csv> gpurun_out/pmc_order.json

SQ_VALU_MFMA_BUSY_CYCLES sums, over the chip's 1024 SIMDs, the cycles their matrix core is busy (MI355X_MICROARCH.md:
32 per v_mfma_f32_32x32x16_bf16, i.e. 512 MAC per cycle and SIMD; the kernels use 16x16x32 = 16 cycles).
utilisation = busy cycles / (1024 SIMDs x launch duration x 2.4 GHz) -- the fraction of the 2.5 PFLOP/s dense bf16
peak (= 1024 SIMDs x 1024 FLOP/cycle x 2.4 GHz) the launch reaches while it runs, duration from the kernel trace of
the SAME run (inflated ~15 % by the counter collection; bench.py's roofline uses un-instrumented HIP-event times).
`expected` = useful MACs / 512 (what the counter would read without tile padding).  GRBM_GUI_ACTIVE is printed raw: it
aggregates several XCD-level instances and is not used as the denominator."""
import csv
import json
import sys


def counter(path, name):
    rows = [r for r in csv.DictReader(open(path)) if r.get("Counter_Name") == name and "gemm" in r.get("Kernel_Name", "")]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return [float(r["Counter_Value"]) for r in rows]


def durations(path):
    rows = [r for r in csv.DictReader(open(path)) if "gemm" in r.get("Kernel_Name", "")]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]


def main(counter_csv, trace_csv, order_json, out_json="profiles/r01i_pmc_mfma_util.json"):
    order = json.load(open(order_json))
    busy, active, dur = counter(counter_csv, "SQ_VALU_MFMA_BUSY_CYCLES"), counter(counter_csv, "GRBM_GUI_ACTIVE"), durations(trace_csv)
    assert len(busy) == len(order) == len(active) == len(dur), (len(busy), len(active), len(dur), len(order))
    agg = {}
    for o, b, a, d in zip(order, busy, active, dur):
        k = (o["name"], o["M"], o["N"], o["K"])
        g = agg.setdefault(k, dict(n=0, busy=0.0, active=0.0, ns=0.0, flops=o["flops"]))
        g["n"] += 1
        g["busy"] += b
        g["active"] += a
        g["ns"] += d
    out = []
    print("%-8s %6s %5s %5s | %12s %12s %10s %9s | %s" % ("gemm", "M", "N", "K", "MFMA busy", "expected", "GRBM active", "us (pmc)", "MFMA utilisation"))
    for (name, M, N, K), g in agg.items():
        b, a, ns = g["busy"] / g["n"], g["active"] / g["n"], g["ns"] / g["n"]
        exp = g["flops"] / 2 / 512
        util = b / (1024.0 * ns * 2.4)
        print("%-8s %6d %5d %5d | %12.0f %12.0f %10.0f %9.2f | %.4f" % (name, M, N, K, b, exp, a, ns / 1e3, util))
        out.append(dict(name=name, M=M, N=N, K=K, mfma_busy_cycles=b, expected_cycles=exp, gpu_active_cycles=a,
                        duration_us_under_pmc=ns / 1e3, mfma_utilisation=util))
    json.dump(out, open(out_json, "w"), indent=1)


if __name__ == "__main__":
    main(*sys.argv[1:])
