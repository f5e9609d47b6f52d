#!/usr/bin/env python3
"""Turn a rocprofv3 `--kernel-trace --stats --output-format csv` directory into
the markdown summary kept under profiles/ (top kernels by total time)."""
import csv
import glob
import sys


def main(src, title):
    f = glob.glob(f'{src}/**/*_kernel_stats.csv', recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    print(f'# {title}\n')
    print('| kernel | calls | total ms | avg us | min us | max us | % |')
    print('|---|---:|---:|---:|---:|---:|---:|')
    for r in rows[:12]:
        name = r['Name'].replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '')
        print(f"| `{name}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.3f} | "
              f"{float(r['AverageNs']) / 1e3:.1f} | {float(r['MinNs']) / 1e3:.1f} | "
              f"{float(r['MaxNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} |")


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else 'rocprofv3 kernel stats')
