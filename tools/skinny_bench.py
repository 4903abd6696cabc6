#!/usr/bin/env python3
"""17-128-row q_gemm (qgemm_skinny.hip) by KERNEL time: run under rocprofv3 --kernel-trace, then `--parse trace.csv meta.json`.
Per (7B linear, M): ours = stage_rows_kernel + qgemm_skinny_kernel, the reference's method = reconstruct_kernel + the library GEMM
(q_gemm.cu:243-263).  Event timing of single calls (tools/prefill_bench.py) is host-bound at these sizes (two launches per call)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

R4 = ([4], [1.0], 128)
SHAPES = [(4096, 4096, ([5, 4], [0.1, 0.9], 128)), (4096, 11008, R4), (11008, 4096, ([8, 4], [0.05, 0.95], [32, 128]))]
MS = (17, 32, 64, 96, 128)
REPS, WARM = 10, 200


def run():
    import torch
    from exllamav2_amd.ext import ext_c as ext, none_tensor
    from exllamav2_amd.synth import synth_linear
    cases = []
    for k, n, rec in SHAPES:
        gen = torch.Generator(device="cuda"); gen.manual_seed(0)
        w = synth_linear(k, n, rec, "cuda", gen)
        h = ext.make_q_matrix_from_dict(w, none_tensor)
        wd = torch.empty((k, n), device="cuda", dtype=torch.float16)
        for m in MS:
            a = torch.randn((m, k), device="cuda", dtype=torch.float16)
            c = torch.empty((m, n), device="cuda", dtype=torch.float16)
            for _ in range(WARM): ext.gemm_half_q_half(a, h, c)
            torch.cuda.synchronize()
            for _ in range(REPS): ext.gemm_half_q_half(a, h, c)
            torch.cuda.synchronize()
            for _ in range(REPS):
                ext.reconstruct(h, wd); torch.matmul(a, wd, out=c)
            torch.cuda.synchronize()
            cases.append({"k": k, "n": n, "m": m, "bytes": int(ext.q_matrix_info(h)["bytes"])})
        ext.free_q_matrix(h)
    print(json.dumps({"cases": cases, "reps": REPS, "warm": WARM}))


def parse(trace, meta):
    import csv
    meta = json.loads(open(meta).read().strip().splitlines()[-1])
    rows = sorted(csv.DictReader(open(trace)), key=lambda r: int(r["Start_Timestamp"]))
    dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0
    sk = [r for r in rows if "qgemm_skinny" in r["Kernel_Name"]]
    st = [r for r in rows if "stage_rows_kernel" in r["Kernel_Name"]]
    rc = [r for r in rows if "reconstruct_kernel" in r["Kernel_Name"]]
    gm = [r for r in rows if r["Kernel_Name"].startswith("Cijk")]
    per = meta["warm"] + meta["reps"]
    med = lambda v: sorted(v)[len(v) // 2]
    for i, c in enumerate(meta["cases"]):
        s = sk[i * per + meta["warm"]:(i + 1) * per]; t = st[i * per + meta["warm"]:(i + 1) * per]
        r = rc[i * meta["reps"]:(i + 1) * meta["reps"]]; g = gm[i * meta["reps"]:(i + 1) * meta["reps"]]
        ours = med([dur(x) for x in s]); pre = med([dur(x) for x in t]); rec = med([dur(x) for x in r]); lib = med([dur(x) for x in g])
        flops = 2.0 * c["m"] * c["k"] * c["n"]
        wbytes = c["bytes"]
        out = dict(c); out["weight_bytes"] = out.pop("bytes")
        out.update({"skinny_us": round(ours, 2), "row_prepass_us": round(pre, 2), "grid": [int(s[0]["Grid_Size_X"]) // 256, int(s[0]["Grid_Size_Y"])],
                    "TFLOPs": round(flops / (ours + pre) / 1e6, 1), "weight_GBps": round(wbytes / ours / 1e3, 0),
                    "reconstruct_us": round(rec, 2), "library_gemm_us": round(lib, 2), "speedup_vs_reconstruct_plus_gemm": round((rec + lib) / (ours + pre), 2)})
        print(json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--parse": parse(sys.argv[2], sys.argv[3])
    else: run()
