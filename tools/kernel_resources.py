"""Static facts about the kernels libmpcbatch.so actually ships, read from the gfx950 code object inside the library (no GPU, no
recompilation): registers, scratch and LDS of every kernel, its instruction classes, and whether anything but kernels was emitted.
    python tools/kernel_resources.py [path/to/libmpcbatch.so]      prints the table kept in profiles/
Used by tests/test_kernel_isa.py (the deterministic guard asked for after the round-2 wrong-code incidents)."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT_SO = os.path.join(ROOT, "mpc_motion_planning_amd", "lib", "libmpcbatch.so")


def _run(*cmd):
    return subprocess.run(cmd, check=True, capture_output=True, text=True).stdout


def code_object(so=DEFAULT_SO, workdir=None):
    """Extract the gfx950 code object from the library's .hip_fatbin section; returns its path."""
    workdir = workdir or tempfile.mkdtemp(prefix="mpcb_co_")
    fat = os.path.join(workdir, "fat.bin"); co = os.path.join(workdir, "dev.co")
    _run(os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", so, fat)
    _run(os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
         "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co)
    return co


def demangle(names):
    out = _run("c++filt", *names).splitlines()
    return [re.sub(r"\(anonymous namespace\)::", "", re.sub(r"^void ", "", o)).split("(")[0] for o in out]


def kernels(so=DEFAULT_SO):
    """{short name: dict(vgpr, agpr, sgpr, scratch, lds_static, instr={class: count}, n_instr)} plus the list of non-kernel functions."""
    co = code_object(so)
    notes = _run(os.path.join(LLVM, "llvm-readelf"), "--notes", co)
    meta = {}
    for blk in re.split(r"\n  - \.agpr_count:", notes)[1:]:
        blk = "  - .agpr_count:" + blk
        g = lambda key: re.search(r"\.%s:\s+(\S+)" % key, blk)   # noqa: E731
        meta[g("name").group(1)] = dict(agpr=int(g("agpr_count").group(1)), vgpr=int(g("vgpr_count").group(1)), sgpr=int(g("sgpr_count").group(1)),
                                        scratch=int(g("private_segment_fixed_size").group(1)), lds_static=int(g("group_segment_fixed_size").group(1)),
                                        wg_max=int(g("max_flat_workgroup_size").group(1)))
    syms = _run(os.path.join(LLVM, "llvm-readelf"), "-s", "-W", co)
    funcs = sorted({l.split()[7] for l in syms.splitlines() if len(l.split()) >= 8 and l.split()[3] == "FUNC"})
    non_kernels = [f for f in funcs if f not in meta]
    dis = _run(os.path.join(LLVM, "llvm-objdump"), "-d", co)
    cur = None
    classes = ("flat_", "scratch_", "ds_", "global_", "buffer_", "v_mfma", "s_waitcnt", "s_barrier")
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            cur = meta.get(m.group(1))
            if cur is not None:
                cur["instr"] = {}; cur["n_instr"] = 0
            continue
        if cur is None:
            continue
        t = line.strip().split()
        if not t or t[0].startswith("//"):
            continue
        cur["n_instr"] += 1
        for c in classes:
            if t[0].startswith(c):
                cur["instr"][c] = cur["instr"].get(c, 0) + 1
    short = demangle(list(meta))
    return {s: meta[n] for s, n in zip(short, meta)}, demangle(non_kernels) if non_kernels else []


def table(so=DEFAULT_SO):
    ks, non = kernels(so)
    lines = ["# kernel                         VGPR AGPR scratch_B  instr   ds_  scratch_  flat_  global_"]
    for name in sorted(ks):
        k = ks[name]; i = k.get("instr", {})
        lines.append("%-32s %4d %4d %8d %7d %5d %8d %6d %7d" % (name, k["vgpr"], k["agpr"], k["scratch"], k.get("n_instr", 0), i.get("ds_", 0),
                                                            i.get("scratch_", 0), i.get("flat_", 0), i.get("global_", 0)))
    lines.append("# functions that are not kernels (the solve functions must be inlined): %s" % (", ".join(non) if non else "none"))
    return "\n".join(lines)


if __name__ == "__main__":
    print(table(sys.argv[1] if len(sys.argv) > 1 else DEFAULT_SO))
