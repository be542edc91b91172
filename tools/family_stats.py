"""Folds a rocprofv3 kernel_stats.csv (--kernel-trace --stats, csv output) into the layer families of the
per-layer profile: calls, total and average duration per family.
usage: python tools/family_stats.py kernel_stats.csv"""
import collections
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_family import family          # noqa: E402


def main():
    fam = collections.defaultdict(lambda: [0, 0.0])
    total = 0.0
    for r in csv.DictReader(open(sys.argv[1])):
        f = family(r['Name']) or 'other'
        fam[f][0] += int(r['Calls'])
        fam[f][1] += float(r['TotalDurationNs'])
        total += float(r['TotalDurationNs'])
    print(f'{"family":28s} {"calls":>7s} {"total ms":>10s} {"avg us":>9s} {"share":>6s}')
    for k, (n, ns) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        print(f'{k:28s} {n:7d} {ns / 1e6:10.3f} {ns / n / 1e3:9.2f} {100 * ns / total:5.1f}%')


if __name__ == '__main__':
    main()
