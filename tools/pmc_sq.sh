#!/bin/bash
# SQ counter pass over bench.py -> gpurun_out/<tag>_pmc_sq_gemm.txt (means per dispatch for the encoder's GEMM shapes and the attention kernel)
tag=${1:-pmcsq}
export TMPDIR=/tmp
out=gpurun_out/_pmcsq_$tag
rm -rf $out; mkdir -p $out
CNT="SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES"
rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $out -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_pmcsq.log 2>&1 || exit 1
python3 - "$out" "$CNT" > gpurun_out/${tag}_pmc_sq_gemm.txt <<'PY'
import collections, csv, glob, os, sys
out, cnt = sys.argv[1], sys.argv[2]
cc = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)[0]
kt = glob.glob(os.path.join(out, "**", "*kernel_trace.csv"), recursive=True)[0]
dur = collections.defaultdict(list)
for r in csv.DictReader(open(kt)):
    dur[(r["Kernel_Name"], r["Grid_Size_X"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(cc)):
    acc[(r["Kernel_Name"], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
print(f"rocprofv3 --pmc {cnt} --kernel-trace -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline   (means per dispatch; SQ_WAVE_CYCLES etc.")
print("count quad-cycles summed over waves, SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs)\n")
want = {"1155072": "QKV (LN consumer, 16-bit out)", "1540096": "MLP-up (LN consumer + GELU, 16-bit out)", "385024": "out-proj / MLP-down (f32 out + 16-bit copy)", "1572864": "attention"}
for (name, grid), c in sorted(acc.items(), key=lambda kv: kv[0][1]):
    if grid not in want or not ("gemm_pp" in name or "attention" in name):
        continue
    d = sorted(dur.get((name, grid), [0.0]))
    print(f"{want[grid]} grid {grid}: {len(c['SQ_WAVE_CYCLES'])} dispatches, median {d[len(d)//2]:.1f} us")
    wc = sum(c["SQ_WAVE_CYCLES"]) / len(c["SQ_WAVE_CYCLES"])
    for k in sorted(c):
        m = sum(c[k]) / len(c[k])
        print(f"   {k:28s} {m:14.0f}   {m / wc:.3f} of SQ_WAVE_CYCLES")
    mf = sum(c["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(c["SQ_VALU_MFMA_BUSY_CYCLES"])
    t = d[len(d)//2] * 1e-6
    for ghz in (1.8, 2.0):
        print(f"   MFMA busy share of SIMD time at {ghz} GHz: {mf / (1024 * t * ghz * 1e9):.3f}")
    print()
PY
rm -rf $out
cat gpurun_out/${tag}_pmc_sq_gemm.txt
