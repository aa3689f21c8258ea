"""stdin: the JSON list tools/bench_field_ops.py / bench_raymarching.py prints; argv[1]: substring of the kernel names to show."""
import json, sys
rows = json.loads(sys.stdin.read().strip().splitlines()[-1])
print("  ".join(f"{r['kernel'].split('[')[-1][:28]}={r['ms']:.4f}" for r in rows if sys.argv[1] in r["kernel"]))
