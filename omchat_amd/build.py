"""Build libomchat_hip.so for gfx950 with hipcc (in-tree: omchat_amd/lib/).  `python -m omchat_amd.build`."""
import os, subprocess, sys, hashlib, concurrent.futures as cf

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libomchat_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
SOURCES = ["gemm.hip", "gemv.hip", "attention.hip", "elementwise.hip", "preproc.hip", "model.hip", "capi.hip", "comm.hip"]
# measured-negative experiment kernels (one-launch decode layer, fused attention + o_proj): NOT part of the product library -- compiled only
# into the `--twin ... -DOMCHAT_EXPERIMENTS=1` build that the experiment tests and tools load through OMCHAT_LIB
EXPERIMENT_SOURCES = ["experiments/fused_decode.hip", "experiments/decode_layer.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-I" + SRC]


def _stamp(path):
    h = hashlib.sha1()
    files = [f for f in sorted(os.listdir(SRC)) if os.path.isfile(os.path.join(SRC, f))]
    files += ["experiments/" + f for f in sorted(os.listdir(os.path.join(SRC, "experiments")))] + ["../../include/omchat_hip.h"]
    for f in files:
        with open(os.path.join(SRC, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def build(force=False, verbose=True, extra_flags=(), out_dir=None):
    """extra_flags / out_dir: an A/B twin of the same ABI built with other -D switches into another directory (load it with OMCHAT_LIB=...;
    `python -m omchat_amd.build --twin ab_lib/f8_nonscaled -DOMCHAT_F8_SCALED=0`)"""
    libdir = out_dir or LIBDIR
    lib = os.path.join(libdir, "libomchat_hip.so")
    os.makedirs(libdir, exist_ok=True)
    stamp_file = os.path.join(libdir, "stamp")
    stamp = _stamp(SRC) + " ".join(extra_flags)
    if not force and os.path.exists(lib) and os.path.exists(stamp_file) and open(stamp_file).read() == stamp:
        return lib
    objs = []

    sources = SOURCES + (EXPERIMENT_SOURCES if "-DOMCHAT_EXPERIMENTS=1" in extra_flags else [])

    def cc(src):
        obj = os.path.join(libdir, os.path.basename(src).replace(".hip", ".o"))
        cmd = [HIPCC, *FLAGS, *extra_flags, "-c", os.path.join(SRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr[-4000:]}")
        return obj

    with cf.ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 2)) as ex:
        objs = list(ex.map(cc, sources))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs, "-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib",
           "-Wl,-z,defs"]      # an undefined symbol fails the build here, not the first dlopen on the GPU box
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stderr[-4000:])
    with open(stamp_file, "w") as f:
        f.write(stamp)
    if verbose:
        print(f"built {lib} ({os.path.getsize(lib)/1e6:.1f} MB)")
    return lib


if __name__ == "__main__":
    if "--twin" in sys.argv:
        i = sys.argv.index("--twin")
        build(force=True, extra_flags=tuple(a for a in sys.argv[i + 2:] if a.startswith("-D")), out_dir=os.path.abspath(sys.argv[i + 1]))
    else:
        build(force="--force" in sys.argv)
