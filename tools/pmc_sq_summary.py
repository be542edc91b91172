"""Fold one rocprofv3 --pmc pass of SQ counters (csv output) into per-kernel-family utilisation.
Per launch: duration from the dispatch timestamps; matrix-core utilisation = SQ_VALU_MFMA_BUSY_CYCLES
(shader cycles, summed over SIMDs) / (SIMDs x duration x clock); vector-issue utilisation =
4 x SQ_ACTIVE_INST_VALU (quad-cycles) / (SIMDs x duration x clock); occupancy = 4 x SQ_WAVE_CYCLES /
(SIMDs x duration x clock) waves per SIMD.  Durations under counter collection are longer than
unprofiled ones, so the ratios are lower bounds.
usage: python tools/pmc_sq_summary.py counter_collection.csv out.json [clock_ghz] [simds]"""
import collections
import csv
import json
import re
import sys


sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from kernel_family import family          # noqa: E402


def main():
    path, out = sys.argv[1], sys.argv[2]
    ghz = float(sys.argv[3]) if len(sys.argv) > 3 else 2.4
    simds = int(sys.argv[4]) if len(sys.argv) > 4 else 1024
    disp = {}
    for r in csv.DictReader(open(path)):
        f = family(r['Kernel_Name'])
        if f is None:
            continue
        d = disp.setdefault(r['Dispatch_Id'], dict(family=f, ns=int(r['End_Timestamp']) - int(r['Start_Timestamp']),
                                                   c=collections.defaultdict(float)))
        d['c'][r['Counter_Name']] += float(r['Counter_Value'])
    fam = collections.defaultdict(lambda: dict(launches=0, ns=0.0, c=collections.defaultdict(float)))
    for d in disp.values():
        f = fam[d['family']]
        f['launches'] += 1
        f['ns'] += d['ns']
        for k, v in d['c'].items():
            f['c'][k] += v
    res = {}
    for name, f in sorted(fam.items(), key=lambda kv: -kv[1]['ns']):
        cyc = f['ns'] * ghz * simds           # SIMD-cycles available over all launches
        c = f['c']
        res[name] = dict(
            launches=f['launches'], avg_us_under_counters=f['ns'] / f['launches'] / 1e3,
            mfma_util=c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / cyc,
            valu_issue_util=4.0 * c.get('SQ_ACTIVE_INST_VALU', 0.0) / cyc,
            waves_per_simd=4.0 * c.get('SQ_WAVE_CYCLES', 0.0) / cyc,
            valu_insts_per_launch=c.get('SQ_INSTS_VALU', 0.0) / f['launches'],
            mfma_mops_bf16_per_launch=c.get('SQ_INSTS_VALU_MFMA_MOPS_BF16', 0.0) / f['launches'],
            lds_insts_per_launch=c.get('SQ_INSTS_LDS', 0.0) / f['launches'],
            lds_bank_conflict_cycles_per_launch=c.get('SQ_LDS_BANK_CONFLICT', 0.0) / f['launches'])
    json.dump(dict(note='one rocprofv3 --pmc pass over tools/pmc_target.py; utilisation = busy / (%d SIMDs x duration x %.1f GHz); '
                        'SQ_VALU_MFMA_BUSY_CYCLES in cycles, SQ_ACTIVE_INST_VALU and SQ_WAVE_CYCLES in quad-cycles' % (simds, ghz),
                   families=res), open(out, 'w'), indent=1)
    for k, v in res.items():
        print(f"{k:24s} n={v['launches']:4d} {v['avg_us_under_counters']:9.1f} us  mfma {v['mfma_util']:.3f}  valu {v['valu_issue_util']:.3f}"
              f"  waves/simd {v['waves_per_simd']:.2f}")


if __name__ == '__main__':
    main()
