"""Register / spill / LDS figures of every kernel in libfbr.so (code-object notes).  python tools/kernel_regs.py [substring] [lib]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "flobaroid_amd", "libfbr.so")
llvm = "/opt/rocm/lib/llvm/bin"
with tempfile.TemporaryDirectory() as td:
    # the fat binary sits in the .hip_fatbin section of the shared object
    subprocess.check_call([os.path.join(llvm, "llvm-objcopy"), "--dump-section", f".hip_fatbin={td}/fat", lib])
    subprocess.check_call([os.path.join(llvm, "clang-offload-bundler"), "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={td}/fat",
                           f"--output={td}/co", "--unbundle"])
    notes = subprocess.check_output([os.path.join(llvm, "llvm-readelf"), "--notes", f"{td}/co"], text=True)
pat = re.compile(r"\.agpr_count:\s+(\d+).*?\.group_segment_fixed_size:\s+(\d+).*?\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_count:\s+(\d+).*?\.sgpr_spill_count:\s+(\d+).*?\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)", re.S)
frag = sys.argv[1] if len(sys.argv) > 1 else ""
for m in pat.finditer(notes):
    agpr, lds, name, priv, sgpr, sspill, vgpr, vspill = m.groups()
    try:
        name = subprocess.check_output(["c++filt", name], text=True).strip()
    except Exception:
        pass
    if frag in name:
        print(f"vgpr={vgpr:>3} agpr={agpr:>3} vspill={vspill:>3} sgpr={sgpr:>3} sspill={sspill:>3} scratch={priv:>5}  {name[:110]}")
