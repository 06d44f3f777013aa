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
    ap.add_argument('--sq', help='directory of the SQ counter pass (wave-cycle breakdown) -> <tag>_sq_stalls.csv')
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
    if a.sq:
        sq_stalls(a.tag, a.sq, out)
        print('sq stall summary written')


def sq_stalls(tag, directory, out):
    f = glob.glob(os.path.join(directory, '*', '*_counter_collection.csv'))[0]
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        per[r['Kernel_Name']][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'SQ_WAVE_CYCLES':
            n[r['Kernel_Name']] += 1
    names = ['SQ_ACTIVE_INST_ANY', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_WAIT_INST_LDS', 'SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE', 'SQ_INSTS_VALU']
    with open(os.path.join(out, f'{tag}_sq_stalls.csv'), 'w', newline='') as fh:
        w = csv.writer(fh)
        w.writerow(['kernel', 'launches', 'wave_cycles_per_launch'] + [x + '/SQ_WAVE_CYCLES' for x in names[:4]] +
                   ['SQ_LDS_BANK_CONFLICT/SQ_LDS_IDX_ACTIVE', 'SQ_INSTS_VALU_per_launch'])
        for k in sorted(per, key=lambda k: -per[k]['SQ_WAVE_CYCLES']):
            wc = per[k]['SQ_WAVE_CYCLES']
            if wc <= 0 or wc < 0.002 * max(v['SQ_WAVE_CYCLES'] for v in per.values()):
                continue
            lds = per[k]['SQ_LDS_IDX_ACTIVE']
            w.writerow([k, n[k], f'{wc / max(n[k], 1):.0f}'] + [f'{per[k][x] / wc:.3f}' for x in names[:4]] +
                       [f'{per[k]["SQ_LDS_BANK_CONFLICT"] / lds:.3f}' if lds > 0 else '', f'{per[k]["SQ_INSTS_VALU"] / max(n[k], 1):.0f}'])


if __name__ == '__main__':
    main()
