#!/usr/bin/env python3
"""Print per-kernel stats from a rocprofv3 results .db (kernel-trace)."""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))  # durations in ns
tot = sum(r[2] for r in rows)
print(f"{'kernel':70s} {'calls':>7s} {'total_us':>12s} {'avg_us':>10s} {'pct':>6s}")
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 20]:
    print(f"{r[0][:70]:70s} {r[1]:7d} {r[2]/1e3:12.1f} {r[3]/1e3:10.2f} {r[4]:6.2f}")
print(f"total kernel time {tot/1e6:.3f} ms")
