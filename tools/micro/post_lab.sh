cd /tmp && export TMPDIR=/tmp
for cfg in "4096 0" "1024 0" "4096 1" "1024 1"; do set -- $cfg; export ODAMD_LAB_ORDER_BLOCK=$1; if [ $2 = 1 ]; then export ODAMD_LAB_NOBBOX=1; else unset ODAMD_LAB_NOBBOX; fi
rm -rf /tmp/pp; timeout -k 10 100 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o pp -- python3 $GRAFT_REPO_ROOT/tools/post_probe.py 5 2>&1 | grep "post ms"
python3 - <<PY
import csv,glob
f=glob.glob("/tmp/pp/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "iou_mask_win" in r["Name"] or "nms_scan" in r["Name"] or "area_order" in r["Name"]: print("  $cfg", r["Name"][21:45], r["Calls"], round(float(r["AverageNs"])/1e3,1), "us avg", round(float(r["TotalDurationNs"])/6e6,3), "ms per call")
PY
done
