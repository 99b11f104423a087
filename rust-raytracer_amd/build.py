"""Builds the native libraries in-tree.

    librrt_hip.so   HIP kernels + C ABI (include/rrt.h), cross-compiled for gfx950
    librrt_host.so  C++ host mirror (loader, Camera, Film, render()) + its C facade (include/rrt_host.h)
    rrt_render      command-line host: rrt_render <scene_folder> [out_prefix] [samples]

hipcc cross-compiles without a GPU, so this runs on the CPU-only build machine
and the resulting .so files travel to the GPU box.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
HOST = os.path.join(HERE, "host")

HIP_SOURCES = ["rrt_api.hip", "bvh_build.cpp", "bvh_build_device.hip"]
HIP_HEADERS = ["rrt_device.h", "rrt_threads.h", "rrt_math.h", "rrt_trace.h", "rrt_shade.h", "rrt_kernels.h"]
HOST_SOURCES = ["yaml_lite.cpp", "obj_loader.cpp", "scene.cpp", "camera_film.cpp", "renderer.cpp", "capi.cpp"]
HOST_HEADERS = ["yaml_lite.h", "obj_loader.h", "rrt_host.hpp"]


def kernel_fingerprint():
    """SHA-256 over the sources the device kernels and the hierarchy they walk are compiled from.  A profile
    summary records it; a reader can tell whether the figures still describe the code in the tree."""
    import hashlib
    h = hashlib.sha256()
    for name in sorted(HIP_SOURCES + HIP_HEADERS):
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read() + b"\0")
    h.update(" ".join(HIP_FP_FLAGS + HIP_CODEGEN_FLAGS).encode())  # the flags shape the kernels as much as the sources do
    return h.hexdigest()


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


# Floating point: no contraction the source does not spell out (the same ray must round the same way in
# every kernel that traces it; the fused operations that matter are written as fmaf), and hardware
# reciprocal / square root (1 ulp) instead of the correctly rounded sequences, which were a fifth of
# the shading kernel's instructions; f32 denormals flush to zero, without which every one of those
# reciprocals and roots is wrapped in a frexp / ldexp rescaling (another fifth).
HIP_FP_FLAGS = ["-ffp-contract=off", "-fno-hip-fp32-correctly-rounded-divide-sqrt", "-fgpu-flush-denormals-to-zero"]
# Code generation: no SLP vectorisation.  On gfx950 the vectoriser pairs independent f32 operations into
# v_pk_mul / v_pk_add / v_pk_fma_f32, which need their operands in adjacent registers; in these kernels (long
# scalar-per-lane arithmetic under register pressure) the moves that line the operands up cost more than the
# paired instruction saves.  Same arithmetic, same results; measured on the bench frame: 84.0 -> 79.3 ms
# (shade 16.3 -> 14.3, extend 32.1 -> 30.6, primary 17.0 -> 15.8 ms).
HIP_CODEGEN_FLAGS = ["-fno-slp-vectorize"]


def _run(cmd, cwd):
    print("+", " ".join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=cwd)


def build_hip(force=False, extra_flags=()):
    out = os.path.join(HERE, "librrt_hip.so")
    deps = [os.path.join(CSRC, f) for f in HIP_SOURCES + HIP_HEADERS] + [os.path.join(ROOT, "include", "rrt.h")]
    if force or _newer(out, deps):
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        _run([hipcc, "--offload-arch=gfx950", "-O3", *HIP_FP_FLAGS, *HIP_CODEGEN_FLAGS, "-std=c++17", "-fPIC", "-shared", "-pthread", *extra_flags,
              "-o", out, *HIP_SOURCES, "-ldl"], CSRC)
    return out


def build_host(force=False):
    out = os.path.join(HERE, "librrt_host.so")
    deps = [os.path.join(HOST, f) for f in HOST_SOURCES + HOST_HEADERS]
    deps += [os.path.join(ROOT, "include", f) for f in ("rrt.h", "rrt_host.h")]
    deps.append(os.path.join(HERE, "librrt_hip.so"))
    if force or _newer(out, deps):
        _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-Wall", "-Wextra", "-o", out, *HOST_SOURCES,
              "-L" + HERE, "-lrrt_hip", "-Wl,-rpath,$ORIGIN"], HOST)
    cli = os.path.join(HERE, "rrt_render")
    if force or _newer(cli, [os.path.join(HOST, "main.cpp"), out]):
        _run(["g++", "-O2", "-std=c++17", "-pthread", "-Wall", "-Wextra", "-o", cli, "main.cpp", "-L" + HERE,
              "-lrrt_host", "-lrrt_hip", "-Wl,-rpath,$ORIGIN"], HOST)
    return out


def build_all(force=False):
    build_hip(force)
    build_host(force)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
