"""Merge TunableOp result files into sug_amd/tuning/tunableop_gfx950.csv (entries of the base file win on a duplicate key; the
Validator lines must agree).  usage: python tools/merge_tunable.py EXTRA.csv [EXTRA2.csv ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
base = os.path.join(ROOT, 'sug_amd', 'tuning', 'tunableop_gfx950.csv')
lines = open(base).read().splitlines()
val = [l for l in lines if l.startswith('Validator')]
seen = {tuple(l.split(',')[:2]) for l in lines if not l.startswith('Validator')}
added = 0
for f in sys.argv[1:]:
    ex = open(f).read().splitlines()
    if [l for l in ex if l.startswith('Validator')] != val:
        print('skipping %s: validator lines differ' % f)
        continue
    for l in ex:
        if l.startswith('Validator') or not l.strip():
            continue
        k = tuple(l.split(',')[:2])
        if k not in seen:
            seen.add(k)
            lines.append(l)
            added += 1
open(base, 'w').write('\n'.join(lines) + '\n')
print('added %d entries -> %d lines' % (added, len(lines)))
