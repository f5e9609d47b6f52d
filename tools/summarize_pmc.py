#!/usr/bin/env python3
"""Combine two rocprofv3 `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes (csv,
--kernel-trace) into per-kernel HBM-side bytes per launch.

usage: summarize_pmc.py <fetch_dir> <write_dir> <out.json> [title] > out.md

Units and correction follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section):
both counters are in KB; on gfx950 FETCH_SIZE reports half of the bytes of wide
coalesced reads, so corrected bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024."""
import csv
import glob
import json
import sys
from collections import defaultdict

ALGORITHMIC = {   # bytes per launch of the default bench step, for the table
}


def collect(d, counter):
    per = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(f'{d}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] != counter:
                continue
            name = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '')
            per[name][0] += float(r['Counter_Value'])
            per[name][1] += 1
    return per


def main(fetch_dir, write_dir, out_json, title):
    fe, wr = collect(fetch_dir, 'FETCH_SIZE'), collect(write_dir, 'WRITE_SIZE')
    rec = {}
    print(f'# {title}\n')
    print('| kernel | launches | FETCH_SIZE KB/launch | WRITE_SIZE KB/launch | corrected bytes/launch (2F+W) |')
    print('|---|---:|---:|---:|---:|')
    for name in sorted(fe, key=lambda n: -fe[n][0]):
        if not name.startswith('ssrs::'):
            continue
        f_kb = fe[name][0] / max(fe[name][1], 1)
        w_kb = wr[name][0] / max(wr[name][1], 1) if name in wr else 0.0
        corr = (2 * f_kb + w_kb) * 1024
        rec[name] = {'launches': fe[name][1], 'fetch_kb': f_kb, 'write_kb': w_kb, 'corrected_bytes': corr}
        print(f'| `{name}` | {fe[name][1]} | {f_kb:.0f} | {w_kb:.0f} | {corr / 1e6:.1f} MB |')
    step = next((v for k, v in rec.items() if 'k_step_' in k), None)   # k_step_lean / k_step_tracks
    out = {'k_step_tracks_bytes_per_launch': step['corrected_bytes'] if step else None,
           'source': f'{title}: (2*FETCH_SIZE + WRITE_SIZE)*1024 per launch, gfx950 half-count '
                     'correction on FETCH_SIZE, separate --pmc passes',
           'kernels': rec}
    json.dump(out, open(out_json, 'w'), indent=1)


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else 'rocprofv3 PMC traffic')
