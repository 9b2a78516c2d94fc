from nf_amd.normalizing_flows.layers.coupling_cells import AffineCoupling, PWLin, PWQuad, RectNN  # noqa: F401
