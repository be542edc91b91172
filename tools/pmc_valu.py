"""Per-kernel totals of SQ_INSTS_VALU / MFMA / LDS from a rocprofv3 counter csv, with the VALU issue floor
(2 cycles per wave64 instruction per SIMD, 1024 SIMDs, 2.4 GHz).  usage: pmc_valu.py file.csv STEPS"""
import collections
import csv
import sys

steps = float(sys.argv[2])
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r['Kernel_Name'][:70]][r['Counter_Name']] += float(r['Counter_Value'])
rows = sorted(acc.items(), key=lambda kv: -kv[1].get('SQ_INSTS_VALU', 0))
tot = 0.0
for k, d in rows[:30]:
    v = d.get('SQ_INSTS_VALU', 0) / steps
    us = v * 2 / 1024 / 2400.0
    tot += us
    print(f"{k:70s} valu/step {v / 1e6:8.1f}M  floor {us:7.1f} us  mfma {d.get('SQ_INSTS_MFMA', 0) / steps / 1e6:6.2f}M  lds {d.get('SQ_INSTS_LDS', 0) / steps / 1e6:6.2f}M salu {d.get('SQ_INSTS_SALU', 0) / steps / 1e6:6.1f}M")
print('sum of VALU floors (top 30): %.1f us per step' % tot)
