"""poppunk_amd: MI355X-native core/accessory distance engine for PopPUNK.

Only the hot path lives here:
  pp_sketchlib.queryDatabase  -> HIP kernel 1 (match counts, Jaccard, regression)
  sketchlib.queryDatabase     -> the PopPUNK-facing wrapper with identical surface
  poppunk_refine.*            -> HIP kernel 2 (assignThreshold / edgeThreshold / generateTuples)
  engine                      -> resident databases, fused distance->edge path, multi-GPU bands
"""
__version__ = "0.1.0"
