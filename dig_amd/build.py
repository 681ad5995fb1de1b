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


if __name__ == "__main__":
    build(force="--force" in sys.argv)
