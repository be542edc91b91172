"""Fold two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; csv output) into per-kernel HBM traffic.
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts a 16-byte-per-lane streaming read at
half its bytes, so reads are doubled (the conv kernels read 16 B per lane).
usage: python tools_pmc_summary.py fetch_counter_collection.csv write_counter_collection.csv out.json"""
import collections
import csv
import json
import re
import sys


def load(path):
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        d[r['Kernel_Name']][0] += 1
        d[r['Kernel_Name']][1] += float(r['Counter_Value']) * 1024.0
    return d


sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from kernel_family import family as _family          # noqa: E402


def family(name):
    return _family(name, conv_only=True)


def main():
    f, w = load(sys.argv[1]), load(sys.argv[2])
    kernels, fam = {}, collections.defaultdict(lambda: [0, 0.0, 0.0])
    for k, (n, fb) in f.items():
        wb = w.get(k, [0, 0.0])[1]
        kernels[k[:140]] = {'launches': n, 'read_bytes_per_launch': 2 * fb / n, 'write_bytes_per_launch': wb / n,
                            'fetch_size_raw_bytes_per_launch': fb / n}
        fm = family(k)
        if fm:
            fam[fm][0] += n
            fam[fm][1] += 2 * fb
            fam[fm][2] += wb
    out = {'note': 'reads = 2 x FETCH_SIZE (gfx950 counts 128-byte requests as 64), writes = WRITE_SIZE; per launch',
           'families': {k: {'launches': n, 'traffic_bytes_per_launch': (r + wr) / n, 'read_bytes_per_launch': r / n,
                            'write_bytes_per_launch': wr / n} for k, (n, r, wr) in fam.items()},
           'kernels': kernels}
    json.dump(out, open(sys.argv[3], 'w'), indent=1)
    for k, v in sorted(out['families'].items()):
        print(k, v)


if __name__ == '__main__':
    main()
