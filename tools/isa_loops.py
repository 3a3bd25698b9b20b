"""Per inner loop of a kernel in a .s file: extent, scratch (spill) instructions inside, VALU cost (tools/isa_cost.py weights).
tools/isa_loops.py file.s kernel_substring"""
import re
import subprocess
import sys

lines = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z.*" + re.escape(key) + r".*:", l))
end = next(i for i in range(start, len(lines)) if ".uses_flat_scratch" in lines[i] or lines[i].startswith(".Lfunc_end"))
headers = {}
for i in range(start, end):
    m = re.match(r"^\.(LBB\d+_\d+):", lines[i])
    if m and i + 1 < len(lines) and "Loop Header" in lines[i + 1] + lines[i]:
        headers[m.group(1)[1:]] = i
for h, i in headers.items():
    members = [j for j in range(start, end) if f"Header={h} " in lines[j] or f"Header={h}\t" in lines[j] or lines[j].rstrip().endswith(f"Header={h}")]
    last = max(members) if members else i
    nxt = next((j for j in range(last + 1, end) if re.match(r"^\.LBB\d+_\d+:", lines[j])), end)
    lo = min([i] + members)
    scr = [j + 1 for j in range(lo, nxt) if "scratch_" in lines[j]]
    depth = re.search(r"Depth=(\d)", lines[i] + lines[i + 1])
    out = subprocess.run([sys.executable, __file__.replace("isa_loops", "isa_cost"), sys.argv[1], str(lo + 1), str(nxt)], capture_output=True, text=True).stdout.strip()
    print(f"loop {h} depth {depth.group(1) if depth else '?'}: scratch ops at {scr[:12]}{'...' if len(scr) > 12 else ''}\n   {out}")
