"""Which workload a dispatch belongs to, for the kernels whose grid does not say (one workgroup per slice of the Gaussians:
geometry_hist_kernel, tile_hist_kernel, tile_scatter_kernel, the idle tile_sort_rare_kernel; their dynamic LDS is not in
rocprofv3's LDS column either): the grid of the next forward-blend dispatch in time order — every forward ends in one."""
BLEND = ("tile_blend_forward_kernel", "blend_forward_kernel")
AMBIGUOUS = ("geometry_hist_kernel", "tile_hist_kernel", "tile_scatter_kernel", "tile_sort_rare_kernel", "table_colscan_kernel")


def short(name):
    return name.split("(")[0].replace("scg::", "").replace("void ", "").split("<")[0].strip()


def norm_grid(kernel_name, grid):
    """The forward blend's grid as tiles x 4 quadrant waves x 64 lanes whatever the variant: the dense frames' instantiation
    (round 5: tile_blend_forward_kernel<3584, 2048, 8, 8>) launches EIGHT waves per tile, four of which only sort."""
    if "tile_blend_forward_kernel<" in kernel_name:
        args = kernel_name.split("tile_blend_forward_kernel<")[1].split(">")[0].split(",")
        if len(args) >= 4 and int(args[3]) > 4:
            return grid * 4 // int(args[3])
    return grid


def tags(rows, grid_key):
    """rows: dicts with Kernel_Name, Dispatch_Id, grid_key.  Returns {dispatch id: blend grid of the forward it belongs to}."""
    seen = {}
    for r in rows:
        seen.setdefault(int(r["Dispatch_Id"]), (short(r["Kernel_Name"]), norm_grid(r["Kernel_Name"], int(r[grid_key]))))
    out, pending = {}, []
    for d in sorted(seen):
        name, grid = seen[d]
        if name in AMBIGUOUS:
            pending.append(d)
        elif name in BLEND:
            for p in pending:
                out[p] = grid
            pending = []
    return out
