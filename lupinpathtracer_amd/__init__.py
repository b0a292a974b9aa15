"""lupinpathtracer_amd -- MI355X (gfx950) software-BVH path tracer behind LupinPathTracer's
`lp::pathtrace_scene` surface.

    api     host-side mirror of the reference's call surface (build_pathtrace_resources, PathtraceDesc,
            AccumulationParams, DoubleBufferedTexture, pathtrace_scene, SceneCPU, ...)
    loader  lupin_loader counterparts (built-in Cornell box, Yocto/GL 2.4 scenes, .hdr I/O)
    csrc/   HIP kernels + the C ABI of include/lupin_hip.h, built in-tree as liblupin_hip.so

There is no CPU rendering path in this package.
"""
from . import api, loader  # noqa: F401
from .api import *  # noqa: F401,F403
