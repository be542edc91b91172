set -o pipefail
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/mk && timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/mk -o mk -- python3 $GRAFT_REPO_ROOT/tools/post_probe.py 2 > /tmp/mk.log 2>&1 || { tail -5 /tmp/mk.log; exit 1; }
f=$(find /tmp/mk -name '*kernel_trace.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
sel = [r for r in rows if any(k in r['Kernel_Name'] for k in ('iou_mask_win', 'nms_scan', 'compact_live', 'area_order', 'pack_cols'))]
for r in sel[-35:]:
    name = r['Kernel_Name'].split('(')[0].split('::')[-1][:24]
    print(f"{name:24s} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f} us")
PY
