"""The `s_waitcnt vmcnt(N)` the compiler emitted, per kernel of the built library (checklist item 26: N near 0 in front of a
prefetched value means the software pipeline is not one -- the wait drains every load AND store the wave has in flight).  Disassembles the
gfx950 code objects inside libartgpu.so (art_amd/codeobj.py + llvm-objdump) and prints, per kernel: vector-memory loads / stores, the
number of vmcnt waits and their histogram (vmcnt(0), 1-3, 4-15, 16+).  Runs without a GPU.
python scripts/waitcnt_audit.py [regex on the demangled kernel name] [path/to/libartgpu.so]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from art_amd import codeobj

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
pat = re.compile(sys.argv[1]) if len(sys.argv) > 1 else None
lib = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "art_amd", "libartgpu.so")

rows = []
for co in codeobj.code_objects(lib):
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(co); f.flush()
        asm = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True, check=True).stdout
    cur, stats = None, {}
    for line in asm.split("\n"):
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            cur = m.group(1); stats[cur] = dict(ld=0, st=0, waits=[]); continue
        if cur is None:
            continue
        t = line.strip()
        if re.match(r"(global|buffer|flat|scratch)_load", t): stats[cur]["ld"] += 1
        elif re.match(r"(global|buffer|flat|scratch)_(store|atomic)", t): stats[cur]["st"] += 1
        elif t.startswith("s_waitcnt"):
            m = re.search(r"vmcnt\((\d+)\)", t)
            if m: stats[cur]["waits"].append(int(m.group(1)))
    rows += [(k, v) for k, v in stats.items() if v["ld"] + v["st"] and not k.endswith(".kd")]
names = codeobj.demangle([k for k, _ in rows])
print(f"{'kernel':64s} {'loads':>5s} {'stores':>6s} {'waits':>5s} {'vm(0)':>5s} {'1-3':>4s} {'4-15':>5s} {'16+':>4s}")
for (k, v), n in sorted(zip(rows, names), key=lambda r: r[1]):
    if pat and not pat.search(n):
        continue
    w = v["waits"]
    print(f"{n[:64]:64s} {v['ld']:5d} {v['st']:6d} {len(w):5d} {sum(x == 0 for x in w):5d} {sum(1 <= x <= 3 for x in w):4d} {sum(4 <= x <= 15 for x in w):5d} {sum(x >= 16 for x in w):4d}")
