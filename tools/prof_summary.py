"""Print the kernel table of a rocprofv3 results.db (top_kernels view), optionally as CSV for profiles/."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
out = open(sys.argv[2], 'w') if len(sys.argv) > 2 else None
if out:
    out.write('"Name","Calls","TotalDurationUs","AverageUs","Percentage"\n')
for name, calls, tot, avg, pct in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    if out:
        out.write(f'"{name}",{calls},{tot:.3f},{avg:.3f},{pct:.4f}\n')
    print(f'{avg:12.1f} us x{calls:5d} {pct:6.2f}%  {name[:110]}')
