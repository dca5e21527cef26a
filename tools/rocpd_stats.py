"""Per-step kernel statistics from a rocprofv3 rocpd database (kernel trace).
usage: python tools/rocpd_stats.py results.db [steps_total] [skip_steps] [csv_out]"""
import sqlite3, collections, re, sys

db = sqlite3.connect(sys.argv[1])
steps_total = int(sys.argv[2]) if len(sys.argv) > 2 else 1
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
rows = list(db.execute("select name, start, end from kernels order by start"))
per = len(rows) // steps_total
rows = rows[len(rows) - per * (steps_total - skip):]
steps = steps_total - skip


def short(n):
    n = n.replace('void ', '').replace('(anonymous namespace)::', '')
    m = re.match(r'at::native::(\w+)<.*?at::native::(\w+)', n)
    if m:
        return 'at::%s<%s>' % (m.group(1), m.group(2))
    n = re.sub(r'\(.*', '', n)
    if n.startswith('Cijk'):
        return n[:60]
    return n[:80]


agg = collections.defaultdict(lambda: [0, 0.0])
for n, s, e in rows:
    k = short(n)
    agg[k][0] += 1
    agg[k][1] += (e - s) / 1e3
tot = sum(v[1] for v in agg.values())
cnt = sum(v[0] for v in agg.values())
span = (rows[-1][2] - rows[0][1]) / 1e6
print('steps %d: kernel time %.3f ms/step, %d launches/step, wall span %.3f ms/step' % (steps, tot / steps / 1e3, cnt / steps, span / steps))
lines = ['kernel,calls_per_step,us_per_step,avg_us,percent']
for n, (k, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    lines.append('"%s",%.1f,%.1f,%.2f,%.2f' % (n, k / steps, t / steps, t / k, 100 * t / tot))
if len(sys.argv) > 4:
    open(sys.argv[4], 'w').write('\n'.join(lines) + '\n')
for l in lines[:int(60)]:
    print(l)
