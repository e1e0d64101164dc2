"""catgrasp_amd -- MI355X-native (gfx950 HIP) implementation of CaTGrasp's grasp-candidate scoring
hot path: PointNet grasp-Q / NUNOCS networks, PointNet++ grouping primitives, the per-candidate
input transform and the my_cpp collision filter.  See DESIGN.md."""
__version__ = '0.1.0'
