"""Condense rocprofv3 output under gpurun_out/ into the small, committed summaries under profiles/.

    python tools/summarize_profiles.py <round-tag> --stats gpurun_out/prof_xxx --pmc gpurun_out

Writes profiles/<tag>_kernel_stats.csv (per-kernel time, from --kernel-trace --stats),
profiles/<tag>_pmc.csv and profiles/<tag>_pmc.json (per-kernel FETCH_SIZE / WRITE_SIZE / MFMA busy, from three
separate --pmc passes over the same bench command)."""

import argparse
import collections
import csv
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def agg(path, counter):
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == counter:
            d[r['Kernel_Name']][0] += 1
            d[r['Kernel_Name']][1] += float(r['Counter_Value'])
    return d


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('tag')
    ap.add_argument('--stats')
    ap.add_argument('--pmc')
    ap.add_argument('--pmc-prefix', default='pmc_')
    ap.add_argument('--steps', type=int, default=5, help='steps (warm-up included) the stats run executed')
    a = ap.parse_args()
    out = os.path.join(ROOT, 'profiles')
    os.makedirs(out, exist_ok=True)
    if a.stats:
        f = glob.glob(os.path.join(a.stats, '*', '*_kernel_stats.csv'))[0]
        rows = list(csv.DictReader(open(f)))
        total = sum(float(r['TotalDurationNs']) for r in rows)
        with open(os.path.join(out, f'{a.tag}_kernel_stats.csv'), 'w', newline='') as fh:
            w = csv.writer(fh)
            w.writerow(['kernel', 'calls', 'total_ms', 'avg_us', 'min_us', 'max_us', 'percent', 'ms_per_step'])
            for r in rows:
                t = float(r['TotalDurationNs'])
                if t / total < 2e-4:
                    continue
                w.writerow([r['Name'], r['Calls'], f'{t / 1e6:.3f}', f"{float(r['AverageNs']) / 1e3:.1f}", f"{float(r['MinNs']) / 1e3:.1f}",
                            f"{float(r['MaxNs']) / 1e3:.1f}", f'{100 * t / total:.2f}', f'{t / 1e6 / a.steps:.2f}'])
        print('kernel time total %.1f ms over %d steps' % (total / 1e6, a.steps))
    if a.pmc:
        g = lambda name: glob.glob(os.path.join(a.pmc, name.replace('pmc_', a.pmc_prefix, 1), '*', '*_counter_collection.csv'))[0]
        fetch = agg(g('pmc_FETCH_SIZE'), 'FETCH_SIZE')
        write = agg(g('pmc_WRITE_SIZE'), 'WRITE_SIZE')
        busy = agg(g('pmc_SQ_VALU_MFMA_BUSY_CYCLES'), 'SQ_VALU_MFMA_BUSY_CYCLES')
        gui = agg(g('pmc_SQ_VALU_MFMA_BUSY_CYCLES'), 'GRBM_GUI_ACTIVE')
        summary = {}
        with open(os.path.join(out, f'{a.tag}_pmc.csv'), 'w', newline='') as fh:
            w = csv.writer(fh)
            w.writerow(['kernel', 'launches', 'FETCH_SIZE_KB_per_launch', 'WRITE_SIZE_KB_per_launch', 'hbm_bytes_per_launch_corrected',
                        'mfma_util_percent'])
            for k in sorted(fetch, key=lambda k: -fetch[k][1]):
                n = fetch[k][0]
                if fetch[k][1] / n < 64 and k not in busy:
                    continue
                fk = fetch[k][1] / n
                wk = write[k][1] / max(write[k][0], 1) if k in write else 0.0
                # MI355X_MICROARCH.md, HBM: FETCH_SIZE reports half the bytes of wide coalesced reads on gfx950 -> x2;
                # WRITE_SIZE is exact.  (x2 is an upper bound for the dword-per-lane gathers of the conv B operand.)
                corrected = (2 * fk + wk) * 1024
                # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 256 CUs x 4 SIMDs = 1024 matrix pipes
                util = 100 * busy[k][1] / (gui[k][1] / 8 * 1024) if k in busy and gui[k][1] > 0 else 0.0
                if fk + wk < 1024 and util < 1:
                    continue
                w.writerow([k, n, f'{fk:.0f}', f'{wk:.0f}', f'{corrected:.0f}', f'{util:.1f}'])
                summary[k] = dict(launches=n, fetch_kb=fk, write_kb=wk, hbm_bytes_corrected=corrected, mfma_util_percent=util)
        import subprocess
        commit = subprocess.run(['git', 'rev-parse', '--short', 'HEAD'], cwd=ROOT, stdout=subprocess.PIPE, text=True).stdout.strip()
        summary['_meta'] = dict(commit=commit, command='python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-meter --no-variants',
                                iterations='0 (every phase) and 1 (Gmain + Dmain): `launches` is over these two',
                                note='bench.py reports roofline.traffic from this file only when its own launch count for the two iterations matches')
        json.dump(summary, open(os.path.join(out, f'{a.tag}_pmc.json'), 'w'), indent=1)
        print('pmc summary for %d kernels' % len(summary))


if __name__ == '__main__':
    main()
