"""Package stub so that `pointnet2.utils.pointnet2_utils` resolves to samplenet_amd.compat (see samplenet_amd/compat/__init__.py)."""
