"""Print VGPR/AGPR/SGPR/spill/LDS of kernels in a HIP fat binary (.so): extracts the gfx950 code object from .hip_fatbin
and reads its msgpack metadata note."""
import subprocess, sys, re, os, tempfile
so = sys.argv[1]; pat = sys.argv[2] if len(sys.argv) > 2 else ""
tmp = tempfile.mkdtemp()
fb = os.path.join(tmp, "fb.bin")
subprocess.check_call(["/opt/rocm/lib/llvm/bin/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", so, fb])
data = open(fb, "rb").read()
# clang offload bundle: find embedded ELF for gfx950
idx = [m.start() for m in re.finditer(b"\x7fELF", data)]
for k, i in enumerate(idx):
    end = idx[k + 1] if k + 1 < len(idx) else len(data)
    co = os.path.join(tmp, "co%d.elf" % k)
    open(co, "wb").write(data[i:end])
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    cur = {}
    for line in out.splitlines():
        line = line.strip()
        m = re.match(r"[- ]*\.(\w+):\s+(.*)", line)
        if not m: continue
        key, val = m.group(1), m.group(2)
        if key == "agpr_count": cur = {"agpr": val}
        elif key == "group_segment_fixed_size": cur["lds"] = val
        elif key == "name" and "kernel" in val or key == "name" and val.startswith("_Z"): cur["name"] = val
        elif key == "private_segment_fixed_size": cur["scratch"] = val
        elif key == "sgpr_count": cur["sgpr"] = val
        elif key == "sgpr_spill_count": cur["sspill"] = val
        elif key == "vgpr_count": cur["vgpr"] = val
        elif key == "vgpr_spill_count":
            cur["vspill"] = val
            if pat in cur.get("name", ""):
                print("%s  vgpr %s agpr %s sgpr %s vspill %s sspill %s scratch %s lds %s" % (cur.get("name"), cur.get("vgpr"), cur.get("agpr"), cur.get("sgpr"), cur.get("vspill"), cur.get("sspill"), cur.get("scratch"), cur.get("lds")))
