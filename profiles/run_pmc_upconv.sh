#!/bin/bash
# counters of the folded-upcat kernels, one forward set on a 256x256x512 dense volume (profiles/upconv_ab.py 1): SQ (two passes),
# FETCH_SIZE and WRITE_SIZE in passes of their own (MI355X_MICROARCH.md: HBM bytes = 2 x FETCH_SIZE [KiB units x 1024 ... see
# make_traffic.py for the corrections] + WRITE_SIZE)
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/${1:-pmc_upconv}
mkdir -p $OUT
cd /tmp
ARGS="$R/profiles/upconv_ab.py 1"
rocprofv3 --kernel-trace --output-format csv -d $OUT/sq --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -- python3 $ARGS > $OUT/sq.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/sq2 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU -- python3 $ARGS > $OUT/sq2.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/fetch --pmc FETCH_SIZE -- python3 $ARGS > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/write --pmc WRITE_SIZE -- python3 $ARGS > $OUT/write.log 2>&1
rm -rf $OUT/*/*/*kernel_trace.csv
python3 - <<PY
import csv, collections, glob, re
res = collections.defaultdict(dict)
for sub in ("sq", "sq2", "fetch", "write"):
    fs = glob.glob("$OUT/%s/*/*counter_collection.csv" % sub)
    if not fs:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
    for r in csv.DictReader(open(fs[0])):
        m = re.search(r"(upconv2m?_kernel<\w+>|conv3_zreg_kernel<[^>]+>|deconv2_regw_kernel<[^>]+>|norm_mish_kernel<[^>]+>|norm_mish_pool_rows_kernel<[^>]+>|final_conv_kernel<[^>]+>|stem_mfma_kernel<[^>]+>)", r["Kernel_Name"])
        if not m:
            continue
        agg[m.group(1)][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[m.group(1)].add(r["Dispatch_Id"])
    for k, v in agg.items():
        res[k].update({c: x / len(cnt[k]) for c, x in v.items()}); res[k]["n"] = len(cnt[k])
for k, v in sorted(res.items()):
    wc = v.get("SQ_WAVE_CYCLES", 1); mf = max(v.get("SQ_INSTS_MFMA", 1), 1)
    print(k, "n=%d" % v["n"])
    print("   valu_busy %.3f" % (v.get("SQ_ACTIVE_INST_VALU", 0) / wc), end="")
    print("   mfma_busy %.3f wait_any %.2f wait_inst_any %.2f active %.2f | per MFMA: VALU %.2f SALU %.2f LDS %.2f VMEM %.3f (MFMA %.3g)" % (
        v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (4 * wc), v.get("SQ_WAIT_ANY", 0) / wc, v.get("SQ_WAIT_INST_ANY", 0) / wc, v.get("SQ_ACTIVE_INST_ANY", 0) / wc,
        v.get("SQ_INSTS_VALU", 0) / mf, v.get("SQ_INSTS_SALU", 0) / mf, v.get("SQ_INSTS_LDS", 0) / mf, (v.get("SQ_INSTS_VMEM_RD", 0) + v.get("SQ_INSTS_VMEM_WR", 0)) / mf, mf))
    print("   FETCH_SIZE %.4g  WRITE_SIZE %.4g (raw counter units per dispatch)" % (v.get("FETCH_SIZE", 0), v.get("WRITE_SIZE", 0)))
PY
