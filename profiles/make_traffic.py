"""HBM traffic per launch from rocprofv3 PMC passes: (2 x FETCH_SIZE + WRITE_SIZE) KB -> bytes.

FETCH_SIZE is doubled as MI355X_MICROARCH.md (section HBM) prescribes for wide (16 B/lane) coalesced reads on gfx950;
WRITE_SIZE is exact for 8/16-byte-per-lane streaming stores.

A kernel template serves several layers (conv3_mfma_kernel<PF16,2,16,false> is four different conv layers), so the
counters are attributed to the library's own labels (the names in bench.py's `kernels`) by walking rocprofv3's dispatch
table in dispatch order next to the launch log the library wrote in the same run (DLV_LAUNCH_LOG: label, algorithmic
flops, algorithmic bytes per bracketed launch, one lane so the order is the stream order).

    python make_traffic.py <dir with fetch/ write/ fetch.launches write.launches> <workload> <precision>
"""
import collections
import csv
import glob
import gzip
import json
import os
import sys

root, wl, precision = sys.argv[1], sys.argv[2], sys.argv[3]
# label prefix -> the kernel(s) a bracketed launch of that label dispatches, in order (a|b: either template)
# (the stem is two launches of one template under one bracket: statistics pass, then the pass that writes)
FAMILY = [("conv3_deep_", ["conv3_deep_kernel"]), ("deconv2_deep_", ["deconv2_deep_kernel"]), ("conv3_zreg_", ["conv3_zreg_kernel"]), ("conv3_zmarch_", ["conv3_zmarch_kernel"]), ("conv3_mfma_", ["conv3_mfma_kernel"]),
          ("norm_mish_", ["norm_mish_kernel|norm_mish_pool_rows_kernel"]), ("pool_act_", ["norm_mish_kernel|norm_mish_pool_rows_kernel"]),
          ("upconv2", ["upconv2m_kernel|upconv2_kernel"]), ("deconv2_mfma_", ["deconv2_rows_kernel|deconv2_regw_kernel|deconv2_wst_kernel"]),
          ("stem_mfma_", ["stem_mfma_kernel", "stem_mfma_kernel"]), ("final_conv_", ["final_conv_kernel"]), ("erode_x_", ["erode_x_kernel|erode_x_bits_kernel"]),
          ("erode_y_", ["erode_y_kernel"]), ("erode_xy_", ["erode_xy_kernel"]), ("erode_z_", ["erode_z_final_kernel|erode_z_shift_kernel"]), ("window_max_", ["cell_max_kernel|window_max_kernel"]),
          ("skip_fill_", ["fill_add_kernel"])]


def family(label):
    for pre, ks in FAMILY:
        if label.startswith(pre):
            return ks
    return None


def base(kernel_name):
    k = kernel_name.replace("void ", "").replace("(anonymous namespace)::", "")
    return k.split("<")[0].split("(")[0].strip()


def load(sub, counter):
    f = glob.glob(os.path.join(root, sub, "*", "*counter_collection.csv*"))
    fh = gzip.open(f[0], "rt") if f[0].endswith(".gz") else open(f[0])
    rows = [r for r in csv.DictReader(fh) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    log = [ln.rstrip("\n").split("\t") for ln in open(os.path.join(root, sub + ".launches"))]
    tot, cnt, alg = collections.Counter(), collections.Counter(), collections.Counter()
    i = 0
    unmatched = 0
    for label, _flops, nbytes in log:
        ks = family(label)
        if ks is None:
            continue
        for want in ks:
            while i < len(rows) and base(rows[i]["Kernel_Name"]) not in want.split("|"):
                i += 1
            if i == len(rows):
                unmatched += 1
                break
            tot[label] += float(rows[i]["Counter_Value"])
            i += 1
        cnt[label] += 1
        alg[label] += float(nbytes)
    return tot, cnt, alg, unmatched


ft, fc, fa, fu = load("fetch", "FETCH_SIZE")
wt, wc, _, wu = load("write", "WRITE_SIZE")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from delivr_cfos_amd._lib import build_fingerprint  # noqa: E402  (hashes only: nothing is loaded)

# what the figures belong to: the library that ran (sha256 of the .so and of its sources) and the commit it was built from
# (DLV_GIT_HEAD: .git does not travel to the GPU box, the caller passes `git rev-parse --short HEAD`); bench.py reports a
# traffic figure only when the library it has loaded matches
out = {"workload": wl, "precision": precision, "round": os.environ.get("DLV_BUILD_TAG", "r06"), "git_head": os.environ.get("DLV_GIT_HEAD"),
       "fingerprint": build_fingerprint(),
       "note": "bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) KB * 1024, averaged over the launches of one label; "
               "algorithmic_bytes = the figure the library's DlvProf bracket declares for that launch (DESIGN.md)",
       "unmatched_launches": fu + wu, "kernels": {}}
for k in sorted(ft, key=lambda k: -(2 * ft[k] + wt[k])):
    fetch = 2 * 1024 * ft[k] / fc[k]
    write = 1024 * wt[k] / max(wc[k], 1)
    a = fa[k] / fc[k]
    out["kernels"][k] = {"launches": fc[k], "fetch_bytes_corrected": fetch, "write_bytes": write, "traffic_bytes": fetch + write,
                         "algorithmic_bytes": a, "traffic_over_algorithmic": (fetch + write) / a if a else None}
print(json.dumps(out, indent=1))
