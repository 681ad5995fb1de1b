"""Build the HIP extension (libdig_hip.so) for gfx950 in-tree with hipcc.  No CPU fallback exists."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libdig_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-ffp-contract=fast",
         "-Wno-unused-result"]
# -fno-slp-vectorize: the SLP vectoriser packs adjacent fp32 multiplies / FMAs of the epilogues into v_pk_*_f32 pairs, which need s_nop
# hazard padding and are no faster than two scalar ops on this chip (MI355X_MICROARCH.md: an anti-lever beside MFMAs): 205 -> 44 s_nop in
# the fused-MLP S-wave loop, and 24.28 -> 24.15 ms per step with it on every file (A/B on one box)
FLAGS.append("-fno-slp-vectorize")
EXTRA_FLAGS = {}                                       # per-file extras (none at present)


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def source_hash() -> str:
    """sha256 (16 hex digits) over the kernel sources (csrc/*.hip, csrc/*.h, csrc/*.inc, include/*.h, this file's flags): what a measurement
    file can be stamped with and re-checked against -- the bytes of the built .so differ from build to build of identical sources."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h", ".inc")))
    inc = os.path.join(os.path.dirname(HERE), "include")
    files += sorted(os.path.join(inc, f) for f in os.listdir(inc) if f.endswith(".h"))
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    h.update(repr((FLAGS, sorted(EXTRA_FLAGS.items()))).encode())
    return h.hexdigest()[:16]


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))]
    inc = os.path.join(os.path.dirname(HERE), "include")
    headers += [os.path.join(inc, f) for f in os.listdir(inc) if f.endswith(".h")]
    headers = [h for h in headers if os.path.exists(h)]
    jobs = []
    for s in sources():
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s[:-4] + ".o")
        if force or _stale(obj, [src] + headers):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(os.path.basename(src), []) + ["-I", os.path.join(os.path.dirname(HERE), "include"), "-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr}")
        if verbose:
            print("[dig_amd.build] compiled", os.path.basename(src), flush=True)

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(cc, jobs))
    objs = [os.path.join(objdir, s[:-4] + ".o") for s in sources()]
    if force or jobs or _stale(LIB, objs):
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs,
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stderr)
        if verbose:
            print("[dig_amd.build] linked", LIB, flush=True)
    return LIB


# ---- the hot kernels of the pre-training step (the families on top of profiles/*_kernel_stats.csv): none may use scratch memory
HOT_KERNELS = ("mlp_chain_kernel<1, true, false>", "mlp_chain_kernel<2, true, false>", "mlp_chain_kernel<0, true, false>",
               "wgrad_wide_kernel<3, 7>", "attn_block_kernel<true, 2>", "attn_block_kernel<false, 2>", "attn_bwd_kernel<false, 3, false>", "attn_bwd_kernel<false, 3, true>",
               "gemm_wide_kernel<false, true, 0, 4, 3, 2, 2, false, 64, 2, false>", "gemm_pwide_kernel<4, false, false>")
LLVM_BIN = os.environ.get("DIG_LLVM_BIN", "/opt/rocm/lib/llvm/bin")


def kernel_resources():
    """[{name (demangled), scratch (private_segment_fixed_size, bytes), vgpr, sgpr, hot}] of every kernel in the built objects."""
    import re
    import tempfile
    rows = []
    objdir = os.path.join(LIBDIR, "obj")
    with tempfile.TemporaryDirectory() as tmp:
        for o in sorted(f for f in os.listdir(objdir) if f.endswith(".o")):
            fat, dev = os.path.join(tmp, o + ".fat"), os.path.join(tmp, o + ".co")
            r = subprocess.run([os.path.join(LLVM_BIN, "llvm-objcopy"), f"--dump-section=.hip_fatbin={fat}", os.path.join(objdir, o)], capture_output=True, text=True)
            if r.returncode != 0 or not os.path.exists(fat):
                continue                                               # (a host-only object)
            subprocess.run([os.path.join(LLVM_BIN, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={fat}",
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={dev}"], check=True, capture_output=True)
            notes = subprocess.run([os.path.join(LLVM_BIN, "llvm-readelf"), "--notes", dev], check=True, capture_output=True, text=True).stdout
            for blk in notes.split("- .agpr_count")[1:]:
                f = {k: re.search(rf"\.{k}:\s+(\S+)", blk) for k in ("name", "private_segment_fixed_size", "vgpr_count", "sgpr_count")}
                if not all(f.values()):
                    continue
                mangled = f["name"].group(1)
                try:
                    dm = subprocess.run(["c++filt", mangled], capture_output=True, text=True).stdout.strip() or mangled
                except OSError:
                    dm = mangled
                short = re.sub(r"^void ", "", dm).replace("(anonymous namespace)::", "")
                rows.append({"name": short, "object": o, "scratch": int(f["private_segment_fixed_size"].group(1)),
                             "vgpr": int(f["vgpr_count"].group(1)), "sgpr": int(f["sgpr_count"].group(1)),
                             "hot": any(short.startswith(h) for h in HOT_KERNELS)})
    return rows


def check_scratch(rows=None, raise_on_fail=True):
    """Every name of HOT_KERNELS exists in the library and has private_segment_fixed_size == 0.  Returns the offending rows."""
    rows = kernel_resources() if rows is None else rows
    missing = [h for h in HOT_KERNELS if not any(r["name"].startswith(h) for r in rows)]
    bad = [r for r in rows if r["hot"] and r["scratch"] > 0]
    if raise_on_fail and (bad or missing):
        raise RuntimeError(f"hot kernels with scratch memory: {[(b['name'][:60], b['scratch']) for b in bad]}; not found: {missing}")
    return bad + [{"name": m + " (not found)", "scratch": -1} for m in missing]


if __name__ == "__main__":
    build(force="--force" in sys.argv)
