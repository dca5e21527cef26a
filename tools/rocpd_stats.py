"""Per-step kernel statistics from a rocprofv3 rocpd database (kernel trace).
usage: python tools/rocpd_stats.py results.db [steps_total] [skip_steps] [csv_out] [marker_regex] [markers_per_step]

Steps are delimited by a MARKER kernel that every training step launches a fixed number of times (default: the Adam
update: `adam_chain_kernel`, one launch per step since round 6, or `adam_kernel`, 3 launches per step = optimizer_dis,
optimizer_g, optimizer_c, in older trees / with SUG_ADAM_CHAIN=0 -- the last kernels of a step): a step
is everything after the previous step's last marker up to and including its own.  The last `steps_total - skip_steps`
complete steps are summarised, so `calls_per_step` is integral for every kernel the steps launch identically.  (Round 4
divided the row count by `steps_total`; a step whose kernels are captured into a hipGraph instead of executed has no
rows, so every per-step figure came out ~10 % high.)  steps_total = 1: the whole run is one "step" (tools/prof_cmd.sh)."""
import sqlite3, collections, re, sys

db = sqlite3.connect(sys.argv[1])
steps_total = int(sys.argv[2]) if len(sys.argv) > 2 else 1
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
rows = list(db.execute("select name, start, end from kernels order by start"))
# since round 6 the three updates are ONE launch (adam_chain_kernel, optim.AdamChain); SUG_ADAM_CHAIN=0 runs bring back 3 x adam_kernel
chained = any('adam_chain_kernel' in r[0] for r in rows)
marker = re.compile(sys.argv[5] if len(sys.argv) > 5 else (r'adam_chain_kernel' if chained else r'adam_kernel'))
per_step = int(sys.argv[6]) if len(sys.argv) > 6 else (1 if chained else 3)

if steps_total > 1:
    marks = [i for i, r in enumerate(rows) if marker.search(r[0])]
    if len(marks) < per_step or len(marks) % per_step:
        sys.exit('rocpd_stats: %d launches of the marker /%s/ are not a multiple of %d per step' % (len(marks), marker.pattern, per_step))
    ends = [marks[i + per_step - 1] for i in range(0, len(marks), per_step)]          # index of each step's last kernel
    bounds = [(0 if i == 0 else ends[i - 1] + 1, e + 1) for i, e in enumerate(ends)]
    want = max(steps_total - skip, 1)
    bounds = bounds[-want:] if len(bounds) >= want else bounds[min(skip, len(bounds) - 1):]
    counts = [b - a for a, b in bounds]
    if len(set(counts)) != 1:
        print('rocpd_stats: WARNING: the summarised steps differ in their launch counts: %s' % counts)
    steps = len(bounds)
    rows = rows[bounds[0][0]:bounds[-1][1]]
    print('steps found by marker /%s/ x%d: %d executed steps in the trace, summarising the last %d (%s launches each)'
          % (marker.pattern, per_step, len(ends), steps, sorted(set(counts))))
else:
    steps = 1


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
    cps = k / steps
    lines.append('"%s",%s,%.1f,%.2f,%.2f' % (n, ('%d' % cps) if cps == int(cps) else ('%.2f' % cps), t / steps, t / k, 100 * t / tot))
if len(sys.argv) > 4 and sys.argv[4]:
    open(sys.argv[4], 'w').write('\n'.join(lines) + '\n')
for l in lines[:int(60)]:
    print(l)
