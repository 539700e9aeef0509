"""Helpers shared by the parity tests (oracle <-> HIP layouts)."""
import numpy as np
import scipy.sparse as sp


def oracle_vel_to_interleaved(u, n2):
    """oracle velocity [ux | uy] (2*n2) -> ours (n2, 2)."""
    return np.stack([u[:n2], u[n2:]], axis=1)


def interleaved_to_oracle_vel(u):
    return np.concatenate([u[:, 0], u[:, 1]])


def device_velocity_matrix(rowptr, colidx, A1, idiag1, pos):
    """Rebuild the (unscaled) velocity operator in the oracle's [ux|uy] ordering
    from the device SELL block arrays (A1 is row-scaled by idiag1); `pos` maps
    CSR non-zeros to SELL positions."""
    n2 = rowptr.size - 1
    nnz = rowptr[-1]
    rows = np.repeat(np.arange(n2), np.diff(rowptr))
    cols = colidx[:nnz]
    blocks = []
    for c in range(2):
        row_blocks = []
        for d in range(2):
            vals = A1[pos, 2 * c + d] / idiag1[rows, c]
            row_blocks.append(sp.coo_matrix((vals, (rows, cols)), shape=(n2, n2)))
        blocks.append(row_blocks)
    return sp.bmat(blocks).tocsr()


def device_sym_matrix(rowptr, colidx, vals, sdiag, pos):
    n = rowptr.size - 1
    nnz = rowptr[-1]
    rows = np.repeat(np.arange(n), np.diff(rowptr))
    cols = colidx[:nnz]
    v = vals[pos] * sdiag[rows] * sdiag[cols]
    return sp.coo_matrix((v, (rows, cols)), shape=(n, n)).tocsr()
