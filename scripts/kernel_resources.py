"""Registers / spills / scratch / static LDS of every kernel in the built library (art_amd/codeobj.py), widest first.
python scripts/kernel_resources.py [path/to/libartgpu.so]      (runs without a GPU)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from art_amd import codeobj

lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "art_amd", "libartgpu.so")
t = codeobj.kernel_table(lib)
print(f"{'kernel':72s} {'vgpr':>4s} {'spill':>5s} {'sgpr':>4s} {'scratch':>7s} {'LDS':>6s} {'wg':>4s}")
for name, r in sorted(t.items(), key=lambda kv: -kv[1]["vgprs"]):
    print(f"{name[:72]:72s} {r['vgprs']:4d} {r['vgpr_spills']:5d} {r['sgprs']:4d} {r['scratch_bytes']:7d} {r['static_lds_bytes']:6d} {r['max_workgroup']:4d}")
