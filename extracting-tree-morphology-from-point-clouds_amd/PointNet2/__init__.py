"""Mirror of the reference's ``Modules/PointNet2`` package: pointnet2_utils, blocks, PointNet2."""
