"""After `gpurun -- bash tools/final_batch.sh`: copy the batch's summaries from gpurun_out/ into profiles/ (tracked), stamp the
commit into r2_pmc.json and append the final bench lines to profiles/r2_bench.jsonl.   python tools/collect_final.py <commit> [tag]"""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
commit = sys.argv[1]
tag = sys.argv[2] if len(sys.argv) > 2 else 'r2'
G, P = os.path.join(ROOT, 'gpurun_out'), os.path.join(ROOT, 'profiles')
for f in ['kernel_stats.csv', 'pmc.csv', 'pmc.json', 'sq_stalls.csv']:
    shutil.copy(os.path.join(G, f'{tag}_profiles', f'{tag}_{f}'), os.path.join(P, f'{tag}_{f}'))
d = json.load(open(os.path.join(P, f'{tag}_pmc.json')))
d['_meta']['commit'] = commit
json.dump(d, open(os.path.join(P, f'{tag}_pmc.json'), 'w'), indent=1)
for f in ['conv_microbench.txt', 'hbm_microbench.txt', 'mfma_shape.txt', 'phase_times.txt']:
    src = os.path.join(G, f'{tag}_{f}')
    if os.path.exists(src) and os.path.getsize(src) > 0:
        shutil.copy(src, os.path.join(P, f'{tag}_{f}'))
probe = os.path.join(G, f'{tag}_power_probe.txt')
if os.path.exists(probe) and os.path.getsize(probe) > 0:
    head = open(os.path.join(P, f'{tag}_power_probe.txt')).readline() if os.path.exists(os.path.join(P, f'{tag}_power_probe.txt')) else ''
    body = open(probe).read()
    open(os.path.join(P, f'{tag}_power_probe.txt'), 'w').write((head if head.startswith('#') else '') + body)
stats = sorted(glob.glob(os.path.join(G, f'{tag}_infer512_stats', '*', '*kernel_stats.csv')), key=os.path.getmtime)
if stats:
    shutil.copy(stats[-1], os.path.join(P, f'{tag}_infer512_kernel_stats.csv'))
path = os.path.join(P, f'{tag}_bench.jsonl')
lines = [l for l in open(path).read().rstrip('\n').split('\n') if f'at {commit}' not in l]
runs = {'bench_final': 'default run (with also_measured and cpu_baseline), same box as the kernel stats / PMC passes',
        'bench_bf16_final': '--storage bf16', 'bench_bf16_512_final': '--storage bf16 --train-res 512 --batch-gpu 8 (config 5, one GPU)',
        'infer512_final': '--mode infer --res 512 (config 4)', 'infer256_final': '--mode infer --res 256'}
for f, what in runs.items():
    dd = json.loads(open(os.path.join(G, f'{tag}_{f}.json')).read().strip().splitlines()[-1])
    dd['_run'] = f'FINAL {what} at {commit}'
    lines.append(json.dumps(dd))
    print(f, dd['value'], dd['ms_per_step'], (dd.get('roofline') or {}).get('frac'), (dd.get('roofline') or {}).get('traffic_source'))
open(path, 'w').write('\n'.join(lines) + '\n')

# one line per kernel: where the waves' cycles go (SQ pass) next to the matrix-pipe utilisation (MFMA-busy pass)
import csv
sq = list(csv.DictReader(open(os.path.join(P, f'{tag}_sq_stalls.csv'))))
pmc = json.load(open(os.path.join(P, f'{tag}_pmc.json')))
def _f(v):
    return float(v) if v not in ('', None) else 0.0
with open(os.path.join(P, f'{tag}_pmc_summary.txt'), 'w') as fh:
    for r in sq[:16]:
        k = r['kernel']
        busy = (pmc.get(k) or {}).get('mfma_util_percent') or 0.0
        fh.write('%-96s wave-cycles %.3e issue %.3f wait %.3f issue-stall %.3f lds-stall %.3f bank-conflict/lds %.3f mfma-busy %.3f\n' % (
            k[:95], _f(r['wave_cycles_per_launch']) * _f(r['launches']), _f(r['SQ_ACTIVE_INST_ANY/SQ_WAVE_CYCLES']), _f(r['SQ_WAIT_ANY/SQ_WAVE_CYCLES']),
            _f(r['SQ_WAIT_INST_ANY/SQ_WAVE_CYCLES']), _f(r['SQ_WAIT_INST_LDS/SQ_WAVE_CYCLES']), _f(r['SQ_LDS_BANK_CONFLICT/SQ_LDS_IDX_ACTIVE']), busy / 100.0))
