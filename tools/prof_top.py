"""Top kernels of a rocprofv3 --kernel-trace --stats run stored as a rocpd database: python tools/prof_top.py <results.db> [n]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
print("%-110s %8s %12s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "%"))
for name, calls, total, avg, pct in db.execute("select name, total_calls, total_duration, average, percentage from top_kernels limit %d" % n):
    print("%-110s %8d %12.1f %10.3f %6.2f" % (name[:110], calls, total / 1e3 if total > 1e7 else total, avg, pct))
