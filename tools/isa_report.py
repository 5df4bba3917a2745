#!/usr/bin/env python3
"""Structural report of the gfx950 code objects in t2ms_amd/csrc/*.o: per kernel the resource metadata (VGPRs, scratch,
LDS) and per LOOP (a backward branch) what the hand-synchronised kernels rely on -- which `s_waitcnt vmcnt(N)` it
contains, how many LDS-DMA instructions (`global_load_lds_*`), other vector-memory instructions (loads, stores, scratch
traffic) and MFMAs.  The counted waits of csrc/t2s_attn.hip / t2s_rows.h / t2s_rows16.h are correct only while the
compiler puts NO vector-memory instruction of its own (a spill reload, a sunk global load) between a DMA batch and the
wait that counts it; tests/test_isa_pins.py pins exactly that against a toolchain upgrade.

    python tools/isa_report.py [stem ...]        # e.g. t2s_attn t2s_dit
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
VMEM = re.compile(r"^(global_|buffer_|scratch_|flat_)")


def extract_code_object(obj_path, workdir):
    """The gfx950 code object bundled in a hipcc host object (llvm-objdump --offloading writes next to its input)."""
    local = os.path.join(workdir, os.path.basename(obj_path))
    shutil.copy(obj_path, local)
    subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", local], check=True, capture_output=True)
    cos = [f for f in os.listdir(workdir) if f.startswith(os.path.basename(obj_path) + ".") and "gfx950" in f]
    if len(cos) != 1:
        raise RuntimeError(f"{obj_path}: expected one gfx950 bundle, found {cos}")
    return os.path.join(workdir, cos[0])


def kernel_metadata(co):
    """{mangled name: the kernel's AMDGPU metadata entry (.vgpr_count, .private_segment_fixed_size, ... without the dot)}"""
    import yaml
    txt = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
    doc = txt[txt.index("---"):]
    doc = doc[:doc.index("\n...")] if "\n..." in doc else doc
    meta = yaml.safe_load(doc)
    return {k[".name"]: {key.lstrip("."): v for key, v in k.items() if key != ".args"} for k in meta["amdhsa.kernels"]}


def disassemble(co):
    """{mangled name: [(address, mnemonic, operands)]}"""
    txt = subprocess.run([f"{LLVM}/llvm-objdump", "-d", co], check=True, capture_output=True, text=True).stdout
    out, cur = {}, None
    for line in txt.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            cur = out.setdefault(m.group(1), [])
            continue
        m = re.match(r"^\s+(\S+)\s*(.*?)\s*// ([0-9A-F]+):", line)
        if m and cur is not None:
            cur.append((int(m.group(3), 16), m.group(1), m.group(2)))
    return out


def loops(insts):
    """Backward branches -> [(first index, last index)] (the last index is the branch), innermost first."""
    addr_to_idx = {a: i for i, (a, _, _) in enumerate(insts)}
    found = []
    for i, (a, mn, ops) in enumerate(insts):
        if mn.startswith("s_cbranch") or mn == "s_branch":
            off = int(ops.split()[0])
            if off >= 32768:
                off -= 65536
            target = a + 4 + 4 * off
            if target <= a and target in addr_to_idx:
                found.append((addr_to_idx[target], i))
    return sorted(found, key=lambda r: r[1] - r[0])


def summarize(insts, lo, hi):
    body = insts[lo:hi + 1]
    waits = [int(re.search(r"vmcnt\((\d+)\)", o).group(1)) for _, m, o in body if m == "s_waitcnt" and "vmcnt" in o]
    return {"insts": len(body), "vmcnt_waits": waits,
            "lds_dma": sum(1 for _, m, _ in body if m.startswith("global_load_lds")),
            "vmem_loads": sum(1 for _, m, _ in body if VMEM.match(m) and "load" in m and not m.startswith("global_load_lds")
                              and not m.startswith("scratch_")),
            "vmem_stores": sum(1 for _, m, _ in body if VMEM.match(m) and "store" in m and not m.startswith("scratch_")),
            "scratch": sum(1 for _, m, _ in body if m.startswith("scratch_")),
            "mfma": sum(1 for _, m, _ in body if m.startswith("v_mfma")),
            "barriers": sum(1 for _, m, _ in body if m == "s_barrier")}


def report(stem, workdir):
    co = extract_code_object(os.path.join(REPO, "t2ms_amd", "csrc", stem + ".o"), workdir)
    meta, dis = kernel_metadata(co), disassemble(co)
    out = {}
    for name, insts in dis.items():
        if name not in meta:
            continue
        ls = [dict(summarize(insts, lo, hi), first=lo, last=hi) for lo, hi in loops(insts)]
        out[name] = {"meta": meta[name], "loops": ls, "insts": len(insts),
                     "mfma": sum(1 for _, m, _ in insts if m.startswith("v_mfma")),
                     "scratch": sum(1 for _, m, _ in insts if m.startswith("scratch_"))}
    return out


def demangled(name):
    try:
        return subprocess.run([f"{LLVM}/llvm-cxxfilt", name], check=True, capture_output=True, text=True).stdout.strip()
    except (OSError, subprocess.CalledProcessError):
        return name


if __name__ == "__main__":
    stems = sys.argv[1:] or ["t2s_attn", "t2s_dit"]
    with tempfile.TemporaryDirectory() as wd:
        for stem in stems:
            for name, r in report(stem, wd).items():
                m = r["meta"]
                print(f"{stem}: {demangled(name)}\n    vgpr {m.get('vgpr_count')} agpr {m.get('agpr_count')} sgpr {m.get('sgpr_count')} "
                      f"scratch {m.get('private_segment_fixed_size')} B lds {m.get('group_segment_fixed_size')} B; "
                      f"{r['insts']} instructions, {r['mfma']} MFMA, {r['scratch']} scratch ops")
                for lp in r["loops"]:
                    if lp["mfma"] or lp["lds_dma"] or lp["vmcnt_waits"]:
                        print(f"    loop [{lp['first']}..{lp['last']}] {lp['insts']} insts: mfma {lp['mfma']} lds_dma {lp['lds_dma']} "
                              f"loads {lp['vmem_loads']} stores {lp['vmem_stores']} scratch {lp['scratch']} barriers {lp['barriers']} "
                              f"vmcnt waits {lp['vmcnt_waits']}")


def streaming_loops(r):
    """The innermost loops of a kernel report that hold both LDS-DMA and MFMAs: the hand-counted rings."""
    hot = [lp for lp in r["loops"] if lp["lds_dma"] and lp["mfma"]]
    inner = [a for a in hot if not any(b is not a and b["first"] >= a["first"] and b["last"] <= a["last"] for b in hot)]
    return [(lp["mfma"], lp["lds_dma"], lp["vmem_loads"], lp["vmem_stores"], lp["scratch"], tuple(lp["vmcnt_waits"]))
            for lp in sorted(inner, key=lambda lp: lp["first"])]
