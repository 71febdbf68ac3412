"""Register / LDS use of the library's kernels, read from the code object embedded in smatrix.so (no GPU needed):
python tools/kernel_regs.py [path/to/smatrix.so] [name filter].  Used to check that a change leaves the instantiations of the
headline path (k_apply_agg<2,1,true,false>, k_apply<0,false>) at their SGPR / VGPR counts -- the residency cliffs of DESIGN.md."""
import os, re, subprocess, sys, tempfile
so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "libsmatrix_amd", "lib", "smatrix.so")
flt = sys.argv[2] if len(sys.argv) > 2 else ""
d = open(so, "rb").read()
i = 0
while True:
    i = d.find(b"\x7fELF", i)
    if i < 0: sys.exit("no gfx code object found")
    if int.from_bytes(d[i + 18:i + 20], "little") == 224: break      # EM_AMDGPU
    i += 4
shoff = int.from_bytes(d[i + 40:i + 48], "little"); size = shoff + int.from_bytes(d[i + 58:i + 60], "little") * int.from_bytes(d[i + 60:i + 62], "little")
with tempfile.NamedTemporaryFile(suffix=".co") as f:
    f.write(d[i:i + size]); f.flush()
    notes = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f.name], capture_output=True, text=True).stdout
cur = {}
rows = []
for line in notes.splitlines():
    m = re.match(r"\s+\.(name|sgpr_count|vgpr_count|sgpr_spill_count|vgpr_spill_count|group_segment_fixed_size):\s+(\S+)", line)
    if not m: continue
    if m.group(1) in cur: rows.append(cur); cur = {}
    cur[m.group(1)] = m.group(2)
if cur: rows.append(cur)
for r in rows:
    name = subprocess.run(["c++filt", r.get("name", "?")], capture_output=True, text=True).stdout.strip().split("(")[0]
    if flt in name:
        print("%-48s sgpr %3s (spilled %2s)  vgpr %3s (spilled %s)  lds %s" % (name[-48:], r.get("sgpr_count"), r.get("sgpr_spill_count"), r.get("vgpr_count"), r.get("vgpr_spill_count"), r.get("group_segment_fixed_size")))
