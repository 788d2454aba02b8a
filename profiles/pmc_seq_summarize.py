"""Join the rocprofv3 passes over profiles/pmc_seq.py (see its header): per launch of rg_seq_kernel the kernel-trace
duration, HBM-side traffic (FETCH_SIZE x2: gfx950 counts the 128-B requests of wide coalesced reads as 64 B,
MI355X_MICROARCH.md section HBM; WRITE_SIZE exact; both in KiB) and the MFMA utilisation
(SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x duration x 2.4 GHz); 16 cycles per v_mfma_f32_16x16x32_bf16).
    python profiles/pmc_seq_summarize.py gpurun_out '<json line printed by pmc_seq.py>' > profiles/r03_pmc_seq.txt"""
import csv, glob, json, os, sys


def find(root, sub, pat):
    hits = glob.glob(os.path.join(root, sub, "**", pat), recursive=True)
    return hits[0] if hits else None


def rows(path, name=None):
    out = [r for r in csv.DictReader(open(path)) if ("rg_seq_kernel" in r.get("Kernel_Name", "") or "rg_seq2_kernel" in r.get("Kernel_Name", "")) and (name is None or r.get("Counter_Name") == name)]
    out.sort(key=lambda r: int(r["Dispatch_Id"]))
    return out


def main(root, info_json):
    info = json.loads(info_json)
    n = info["launches"]
    res = dict(info)
    tr = find(root, "seq_stats", "*kernel_trace.csv")
    if tr:
        d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows(tr)][-n:]
        res["avg_us"] = sum(d) / len(d) / 1e3
        res["tflops"] = info["flops_per_launch"] / (res["avg_us"] * 1e-6) / 1e12
    for sub, cname, key, scale in (("seq_fetch", "FETCH_SIZE", "fetch_bytes", 2.0 * 1024), ("seq_write", "WRITE_SIZE", "write_bytes", 1024.0)):
        c = find(root, sub, "*counter_collection.csv")
        if c:
            v = [float(r["Counter_Value"]) for r in rows(c, cname)][-n:]
            res[key] = scale * sum(v) / len(v)
    c, t = find(root, "seq_mfma", "*counter_collection.csv"), find(root, "seq_mfma", "*kernel_trace.csv")
    if c and t:
        busy = [float(r["Counter_Value"]) for r in rows(c, "SQ_VALU_MFMA_BUSY_CYCLES")][-n:]
        dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows(t)][-n:]
        res["mfma_busy_cycles"] = sum(busy) / len(busy)
        res["us_under_pmc"] = sum(dur) / len(dur) / 1e3
        res["mfma_utilisation"] = res["mfma_busy_cycles"] / (1024.0 * res["us_under_pmc"] * 1e3 * 2.4)
        res["expected_busy_cycles"] = info["flops_per_launch"] / 2 / 512 * (48.0 / 43.0)
    print(json.dumps(res, indent=1))
    if len(sys.argv) > 3:
        with open(sys.argv[3], "w") as f:
            json.dump(res, f, indent=1)
    if "fetch_bytes" in res and "write_bytes" in res:
        print("HBM-side traffic per launch %.1f MB (fetch %.1f + write %.1f) against %.1f MB algorithmic: x%.2f; LDS-ring intake %.1f GB per launch"
              % ((res["fetch_bytes"] + res["write_bytes"]) / 1e6, res["fetch_bytes"] / 1e6, res["write_bytes"] / 1e6,
                 info["algorithmic_hbm_bytes"] / 1e6, (res["fetch_bytes"] + res["write_bytes"]) / info["algorithmic_hbm_bytes"], info["lds_ring_bytes"] / 1e9))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
