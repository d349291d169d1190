"""pislam_amd — MI355X-native ORB front-end behind the PiSlam API.

  pislam_amd.frontend : host-side mirror of the reference interface (fastDetect, fastScoreHarris,
                        fastExtract, orbCompute, ... and the batch OrbFrontend) over the C ABI
  pislam_amd.capi     : ctypes binding of libpislam_hip.so (include/pislam_hip.h)
  pislam_amd.build    : in-tree hipcc build for gfx950
  pislam_amd.synth    : deterministic synthetic stacked pyramids (bench / test workload)
  pislam_amd.dist     : one-process-per-GPU sharding + count all-gather (torch.distributed)
"""
__version__ = "0.1.0"
