from nf_amd.normalizing_flows.layers.coupling_cells import PWLin, PWQuad, RectNN  # noqa: F401
