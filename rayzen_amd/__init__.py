"""rayzen_amd -- MI355X-native path-tracing render loop behind RayZen's scene API.

  rayzen_amd.scene     host-side scene assembly (C++ builders in librayzen_host.so)
  rayzen_amd.renderer  the render C-ABI of include/rayzen_hip.h (HIP kernels, librayzen_hip.so)
  rayzen_amd.dist      tile sharding across GPUs + the one reduce
  rayzen_amd.build     compiles both libraries in-tree
"""
__version__ = "0.1"
