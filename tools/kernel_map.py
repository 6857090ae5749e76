"""Which launch of a cycle is which: for every operator of a resident hierarchy the kernels a V-cycle launches on it, keyed
the way rocprofv3's kernel trace shows them (kernel family + epilogue + grid size in workgroups), with the algorithmic bytes
of SURVEY.md 8(d) and the bytes the operator format that runs actually streams.  bench.py --kernel-map writes it next to
the trace; tools/summarize_prof.py joins it with the measured durations into the per-kernel roofline table.  Not product
code."""
import numpy as np


def _vec_bytes(epi, nr, nc, vb=8):
    return {"SET": vb * (nc + nr), "ACC": vb * (nc + 2 * nr), "RESID": vb * (nc + 2 * nr), "SUMSQ": vb * (nc + nr),
            "AXPBY": vb * (nc + 2 * nr), "ACC_AXPBY": vb * (nc + 3 * nr), "JACOBI": vb * (nc + 2 * nr), "JACOBI_B": vb * (nc + 2 * nr)}[epi]


def _op_entry(level, opname, dA, epi, what):
    inf = dA.info()
    nr, nc, nnz = inf["rows"], inf["cols"], inf["nnz"]
    vb = dA.dtype.itemsize
    vec = _vec_bytes(epi, nr, nc, vb)
    alg = (vb + 4) * nnz + 4 * (nr + 1) + vec
    npat, nval = dA.row_patterns(), dA.value_codes()
    masks = dA.row_masks() if npat else {"entries": 0}
    grid = int(inf["row_blocks"])
    if npat and masks["entries"]:
        streamed, form = nr + vec, ("row masks, lattice tiles 64 x 4 x %d (1 byte per row)" % masks["planes_per_lane"]) if (masks["lattice"] and epi != "SUMSQ") else "row masks (1 byte per row)"
        if epi != "SUMSQ":
            grid = int(masks["launch_grid"])
    elif npat:
        streamed, form = nr + vec, "row patterns (1 byte per row)"
    elif nval:
        streamed, form = 3 * nnz + 4 * (nr + 1) + vec, "16-bit columns + 8-bit value codes (3 bytes per entry)"
    else:
        streamed, form = (vb + 2) * nnz + 4 * (nr + 1) + vec, "16-bit columns + values as stored (10 bytes per entry)"
    return {"family": "csr", "epi": epi, "grid": grid, "level": level, "op": opname, "what": what,
            "rows": int(nr), "nnz": int(nnz), "bytes_alg": int(alg), "bytes_streamed": int(streamed), "format": form}


def kernel_map(dml, smoother_kind):
    nlev = len(dml.A)
    mats = list(dml._mats)
    out = []
    k = 0
    for i in range(nlev):
        A = mats[k]; k += 1
        P = R = None
        if i < nlev - 1:
            P, R = mats[k], mats[k + 1]
            k += 2
        if i == 0:
            out.append(_op_entry(i, "A", A, "SUMSQ", "convergence-check norm ||b - A x||"))
        if i < nlev - 1:
            out.append(_op_entry(i, "A", A, "RESID", "residual r = b - A x"))
            out.append(_op_entry(i, "R", R, "SET", "restriction b_c = R r"))
            out.append(_op_entry(i, "P", P, "ACC", "prolongation x += P x_c"))
            inf = A.info()
            n, nnz, vb = inf["rows"], inf["nnz"], A.dtype.itemsize
            if smoother_kind == "gauss_seidel":
                alg = (vb + 4) * nnz + 4 * (n + 1) + 3 * vb * n
                for which, dirn in ((0, "forward"), (1, "backward")):
                    lane, tile, line, lanem = A.lane_info(which), A.tile_info(which), A.line_info(which), A.lanem_info(which)
                    if lanem["rows"] and lanem["launch_grid"]:
                        out.append({"family": "gs_lanem", "grid": int(lanem["launch_grid"]), "level": i, "op": "A",
                                    "what": f"{dirn} Gauss-Seidel sweep (fast order, merged: {lanem['super_levels']} super-levels of <= {lanem['s_max']} dependency levels)",
                                    "rows": int(n), "nnz": int(nnz), "bytes_alg": int(alg), "bytes_streamed": int(lanem["units"] * 64 * (vb + 4) + n * (32 + 4 * vb)),
                                    "format": f"one row per wave, {lanem['units'] / max(1, lanem['rows']):.2f} x 64 operand slots per row, padded", "dependency_levels": int(lanem["super_levels"])})
                    elif line["lines"] and line["launch_grid"]:
                        out.append({"family": "gs_line", "grid": int(line["launch_grid"]), "level": i, "op": "A", "what": f"{dirn} Gauss-Seidel sweep (fast order, line scan: {line['lines']} lines, {line['line_levels']} line levels)",
                                    "rows": int(n), "nnz": int(nnz), "bytes_alg": int(alg), "bytes_streamed": int(line["chunks"] * 64 * (line["slots_per_row"] * (vb + 4) + 2 * vb + 1) + n * 4 * vb),
                                    "format": f"chunks of 64 rows x {line['slots_per_row']} slots", "dependency_levels": int(line["line_levels"])})
                    elif lane["groups"] and lane["launch_grid"]:
                        out.append({"family": "gs_lane", "grid": int(lane["launch_grid"]), "level": i, "op": "A", "what": f"{dirn} Gauss-Seidel sweep (fast order, {lane['lanes_per_row']} lanes per row)",
                                    "rows": int(n), "nnz": int(nnz), "bytes_alg": int(alg), "bytes_streamed": int(lane["entry_slots"] * (vb + 4) + n * (4 + 4 * vb)),
                                    "format": f"{lane['lanes_per_row']} lanes x {lane['slots_per_lane']} slots per row, padded", "dependency_levels": int(inf["gs_levels_fwd"])})
                    elif tile["tiles"]:
                        out.append({"family": "gs_tile", "grid": int(tile["tiles"]), "level": i, "op": "A", "what": f"{dirn} Gauss-Seidel sweep (order-exact, tiled)",
                                    "rows": int(n), "nnz": int(nnz), "bytes_alg": int(alg), "bytes_streamed": None, "format": "step blocks", "dependency_levels": int(inf["gs_levels_fwd"])})
                    else:
                        out.append({"family": "gs_gran", "grid": None, "level": i, "op": "A", "what": f"{dirn} Gauss-Seidel sweep (order-exact)",
                                    "rows": int(n), "nnz": int(nnz), "bytes_alg": int(alg), "bytes_streamed": None, "format": "level-permuted CSR", "dependency_levels": int(inf["gs_levels_fwd"])})
            elif smoother_kind in ("block_gauss_seidel", "block_jacobi") or (A.op.fmt == "bsr" and A.op.blocksize[0] > 1 and smoother_kind in ("gauss_seidel", "jacobi")):
                # square-block levels: SURVEY 8(d) BSR(R x C) bytes = 8 R C nblk + 4 nblk + 4 (n_brow + 1) + vectors (x read, b read, x
                # written) + the inverted diagonal blocks of the block smoothers (8 R^2 per block row)
                bs = int(A.op.blocksize[0])
                nbrow = n // bs
                nblk = nnz // (bs * bs)
                block = smoother_kind.startswith("block_")
                alg = vb * bs * bs * nblk + 4 * nblk + 4 * (nbrow + 1) + 3 * vb * n + (vb * bs * bs * nbrow if block else 0)
                gs = smoother_kind.endswith("gauss_seidel")
                kind = ("BLK_GS" if block else "PNT_GS") if gs else ("BLK_JACOBI" if block else "PNT_JACOBI")
                blanes = [A.lane_info(w) for w in (0, 1)] if (gs and block) else []
                if gs and block and any(l_["groups"] and l_["launch_grid"] for l_ in blanes):
                    # fast order (csrc/pamg_blane.hip): one entry per direction; streamed = the padded block slots (values + block column), the
                    # block row of every slot row, b / Dinv / the result, the hand-off buffer (filled, written, polled) and x
                    for which, dirn in ((0, "forward"), (1, "backward")):
                        l_ = blanes[which]
                        if not (l_["groups"] and l_["launch_grid"]):
                            continue
                        rpw = 64 // l_["lanes_per_row"]
                        out.append({"family": "bsr_lane", "kind": kind, "bs": bs, "grid": int(l_["launch_grid"]), "level": i, "op": "A",
                                    "what": f"{dirn} block Gauss-Seidel sweep on BSR({bs},{bs}) (fast order, {l_['lanes_per_row']} lanes per block row)",
                                    "rows": int(n), "nnz": int(nnz), "bytes_alg": int(alg),
                                    "bytes_streamed": int(l_["entry_slots"] * (vb * bs * bs + 4) + l_["groups"] * rpw * 4 + n * 5 * vb + vb * bs * bs * nbrow),
                                    "format": f"{l_['lanes_per_row']} lanes x {l_['slots_per_lane']} blocks per block row, padded", "dependency_levels": int(inf["gs_levels_fwd"])})
                elif gs:
                    out.append({"family": "bsr_gs", "kind": kind, "bs": bs, "grid": None, "level": i, "op": "A",
                                "what": f"{'block' if block else 'point'} Gauss-Seidel sweep on BSR({bs},{bs}) (order-exact; forward and backward launches)",
                                "rows": int(n), "nnz": int(nnz), "bytes_alg": int(alg), "bytes_streamed": int(alg + 4 * nbrow), "format": f"level-permuted BSR({bs},{bs}) + row ids",
                                "dependency_levels": int(inf["gs_levels_fwd"])})
                else:
                    out.append({"family": "bsr_stream", "kind": kind, "bs": bs, "grid": int(inf["row_blocks"]), "level": i, "op": "A",
                                "what": f"{'block' if block else 'point'} Jacobi sweep on BSR({bs},{bs})", "rows": int(n), "nnz": int(nnz), "bytes_alg": int(alg),
                                "bytes_streamed": int(alg), "format": f"BSR({bs},{bs}) as stored"})
            elif smoother_kind == "jacobi":
                out.append(_op_entry(i, "A", A, "JACOBI", "weighted Jacobi sweep"))
                out.append(_op_entry(i, "A", A, "JACOBI_B", "weighted Jacobi sweep (BSR(1,1) levels)"))
            elif smoother_kind == "chebyshev":
                out.append(_op_entry(i, "A", A, "AXPBY", "Horner step h = c r + A h"))
                out.append(_op_entry(i, "A", A, "ACC_AXPBY", "last Horner step folded with x += h"))
                out.append(_op_entry(i, "A", A, "SET", "first product of the polynomial (x = 0)"))
    return out
