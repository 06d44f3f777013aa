"""Share of a file's code lines that also occur (whitespace-normalised) in a reference file -- the measure the round-1
review used for copied class bodies.  Usage: python tools/copycheck.py <ours> <reference>"""
import re
import sys

def lines(path):
    out = []
    for ln in open(path, errors='replace'):
        ln = re.sub(r'\s+', '', ln.split('#')[0] if '#' in ln and "'#" not in ln else ln)
        if len(ln) > 3 and not ln.startswith(('"""', "'''")):
            out.append(ln)
    return out

ours, ref = lines(sys.argv[1]), set(lines(sys.argv[2]))
hit = [l for l in ours if l in ref]
print(f'{len(hit)} of {len(ours)} code lines ({100 * len(hit) / max(len(ours), 1):.0f} %) also occur in {sys.argv[2]}')
if len(sys.argv) > 3:
    print('\n'.join(sorted(set(hit))))
