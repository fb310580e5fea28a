"""Register / spill / LDS table of the kernels in a built library (from the code object's metadata notes):
    python tools/kernel_regs.py [pistoseg_amd/libpistoseg_hip.so] [name-filter]
Lists vgpr / agpr / sgpr counts, spilled registers, scratch bytes and static LDS per kernel -- the check that a new template instantiation
of a conv kernel did not start spilling (a spill in a consumer wave is a VMEM load with a vmcnt wait: NOTES 7.9)."""
import re
import subprocess
import sys

lib = sys.argv[1] if len(sys.argv) > 1 else "pistoseg_amd/libpistoseg_hip.so"
flt = sys.argv[2] if len(sys.argv) > 2 else ""
tmp = "/tmp/_kregs"
subprocess.run(["rm", "-rf", tmp]); subprocess.run(["mkdir", "-p", tmp])
# the library's .hip_fatbin section holds one offload bundle per translation unit: unbundle each, read its msgpack metadata as text
subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, f"{tmp}/fat.bin"], check=True)
blob = open(f"{tmp}/fat.bin", "rb").read()
magic = b"__CLANG_OFFLOAD_BUNDLE__"
starts = [i for i in range(len(blob)) if blob.startswith(magic, i)]
out = ""
for n, st in enumerate(starts):
    en = starts[n + 1] if n + 1 < len(starts) else len(blob)
    open(f"{tmp}/b{n}.bin", "wb").write(blob[st:en])
    subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--type=o", f"--input={tmp}/b{n}.bin", "--unbundle",
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={tmp}/dev{n}.co"], check=True, capture_output=True)
    out += subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f"{tmp}/dev{n}.co"], capture_output=True, text=True).stdout
kern = []
cur = {}
for ln in out.splitlines():
    m = re.match(r"\s+-?\s*\.(\w+):\s+(.*)", ln)
    if not m:
        continue
    k, v = m.group(1), m.group(2).strip()
    if k == "agpr_count" and cur.get("name"):
        kern.append(cur); cur = {}
    if k in ("name", "vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size"):
        if k == "name" and "name" in cur and "vgpr_count" in cur:
            kern.append(cur); cur = {}
        if k == "name" and ("vgpr_count" not in cur):
            cur["name"] = v
        elif k != "name":
            cur[k] = v
if cur.get("name"):
    kern.append(cur)
seen = set()
for k in kern:
    nm = k.get("name", "?")
    if nm in seen or "vgpr_count" not in k:
        continue
    seen.add(nm)
    dem = subprocess.run(["c++filt", nm], capture_output=True, text=True).stdout.strip().replace("(anonymous namespace)::", "")
    if flt and flt not in dem:
        continue
    print(f"v{k.get('vgpr_count','?'):>4} a{k.get('agpr_count','?'):>4} s{k.get('sgpr_count','?'):>4} vspill {k.get('vgpr_spill_count','0'):>4} sspill {k.get('sgpr_spill_count','0'):>4} "
          f"scratch {k.get('private_segment_fixed_size','0'):>5} lds {k.get('group_segment_fixed_size','0'):>6}  {dem[:150]}")
