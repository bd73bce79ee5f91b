"""MI355X-native Whitted render path (drop-in for renderRayTracing / getFinalColor /
BvhInterface::intersect of Anton-Kalpakchiev/Computer-Graphics-Engine).

Layout: csrc/ = HIP kernels + C ABI (include/rt_whitted.h); host/ = C++ shim with
the reference's call surface; abi.py / capi.py = ctypes binding; scenes.py =
scene containers, fixtures and synthetic scenes; sharding.py = multi-GPU strips.
The package name contains a hyphen: import it with
importlib.import_module("computer-graphics-engine_amd") or via the root-level
alias module `cge_amd`.
"""
