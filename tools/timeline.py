#!/usr/bin/env python
"""Per-stream timeline of one steady-state frame from a rocprofv3 --kernel-trace rocpd database (run_results.db)."""
import collections, sqlite3, sys
db = sys.argv[1]
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
c = sqlite3.connect(db)
names = {r[0]: r[1] for r in c.execute("select id, kernel_name from rocpd_info_kernel_symbol")}
rows = c.execute("select kernel_id, queue_id, stream_id, start, end from rocpd_kernel_dispatch order by start").fetchall()
def nm(k):
    n = names.get(k, str(k)).replace('(anonymous namespace)::', '').replace('eodconv::', '')
    return n.split('(')[0][-48:]
norm = [i for i, r in enumerate(rows) if 'normalize_dirty' in nm(r[0])]
i0, i1 = norm[25], norm[26]
t0, t1 = rows[i0][3], rows[i1][3]
print('frame span ms', (t1 - t0) / 1e6)
fr = [r for r in rows if t0 <= r[3] < t1]
byq = collections.defaultdict(list)
for r in fr:
    byq[r[1]].append(r)
for q, l in byq.items():
    busy = sum(r[4] - r[3] for r in l)
    print('queue', q, 'n', len(l), 'first', round((l[0][3] - t0) / 1e6, 3), 'last_end', round((max(r[4] for r in l) - t0) / 1e6, 3), 'busy ms', round(busy / 1e6, 3))
marks = ('preprocess', 'gather_pool', 'cn_merge', 'mw_apply', 'postprocess', 'det_candidates', 'roi_align', 'mw_unique')
for q, l in byq.items():
    print('--- queue', q)
    for r in l:
        d = (r[4] - r[3]) / 1e3
        if d > thr or any(m in nm(r[0]) for m in marks):
            print(round((r[3] - t0) / 1e6, 3), round(d, 1), nm(r[0]))
