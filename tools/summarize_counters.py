#!/usr/bin/env python3
"""Per-kernel sums of arbitrary rocprofv3 --pmc counters (csv, --kernel-trace) from one or more
pass directories -> markdown table (per dispatch averages).
usage: summarize_counters.py title dir [dir ...] > out.md"""
import csv
import glob
import sys
from collections import defaultdict


def main(title, dirs):
    per = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(lambda: defaultdict(int))
    counters = []
    for d in dirs:
        for f in glob.glob(f'{d}/**/*counter_collection.csv', recursive=True):
            for r in csv.DictReader(open(f)):
                name = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '')
                c = r['Counter_Name']
                if c not in counters:
                    counters.append(c)
                per[name][c] += float(r['Counter_Value'])
                calls[name][c] += 1
    print(f'# {title}\n')
    print('Per-dispatch averages (sum over the dispatches of a kernel / number of dispatches).\n')
    print('| kernel | dispatches | ' + ' | '.join(counters) + ' |')
    print('|---|---:|' + '---:|' * len(counters))
    names = [n for n in per if n.startswith('ssrs::')]
    names.sort(key=lambda n: -max(per[n].values()))
    for n in names:
        k = max(calls[n].values())
        cells = [f'{per[n][c] / max(calls[n][c], 1):.4g}' if c in per[n] else '' for c in counters]
        print(f'| `{n}` | {k} | ' + ' | '.join(cells) + ' |')


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2:])
