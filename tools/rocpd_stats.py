"""Per-kernel statistics (calls, total, average, share) from a rocprofv3 rocpd SQLite database, as CSV on stdout.
usage: python tools/rocpd_stats.py results.db [top_n]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = db.execute("""select s.kernel_name, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), max(d.end - d.start)
                     from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
                     group by s.kernel_name order by 3 desc""").fetchall()
total = sum(r[2] for r in rows)
print("Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage")
for r in rows[:top]:
    print('"%s",%d,%d,%.0f,%d,%d,%.2f' % (r[0][:110], r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / total))
print('"TOTAL",%d,%d,,,,100.00' % (sum(r[1] for r in rows), total))
