"""The pytorch3d names the reference imports (pn_kit.py:10, pointnet_sa_module.py:4, AE.py:7), served
by libpccx.so.  Semantics are the oracle's definitions: PARITY UNPINNED against the
real pytorch3d for tie order / padding."""
