"""torch-CPU restatement of the reference's closest-cylinder projection in its own structure -- TEST / BENCH
INFRASTRUCTURE ONLY, PARITY UNPINNED (Modules/Projection.py imports `fastprogress`, which is not installed).

Same algorithmic structure as Modules/Projection.py:19-144: batches of 1024 points broadcast against all M cylinders with
[1024, M, 3] temporaries, argmin, fancy indexing of the winner.  Used as the cpu_baseline of tools/bench_datapath.py; the
bit-level oracle of the kernel is oracle/pn2_oracle.c (pn2o_cylinder_project)."""
import numpy as np
import torch


def closest_cylinder_batch(points, start, radius, axis_length, axis_unit, IDs, move_points_to_mantle=True):
    points = torch.as_tensor(points, dtype=torch.float32)
    pv = points[:, None, :] - start[None, :, :]                                               # :33
    pl = torch.sum(pv * axis_unit[None, :, :], dim=2, keepdim=True)                           # :36
    plc = torch.clamp(pl, torch.zeros_like(pl), axis_length[None, :, :])                      # :39-40
    ppc = start[None, :, :] + plc * axis_unit[None, :, :]                                     # :41
    prv = points[:, None, :] - ppc                                                            # :44
    dp = torch.sum(prv * axis_unit[None, :, :], dim=2)                                        # :47
    perp = torch.isclose(dp, torch.tensor(0.0), atol=1e-3)                                    # :48
    rej = prv - dp[..., None] * axis_unit[None, :, :]                                         # :51-52
    nr = torch.norm(rej, dim=2, keepdim=True)                                                 # :55
    safe = nr.clone()
    safe[safe < 1e-8] = 1e-8                                                                  # :58-60
    nau = rej / safe                                                                          # :61
    nas = nau * (2 * radius.view(1, -1, 1))                                                   # :64
    ns, ne = ppc - 0.5 * nas, ppc + 0.5 * nas                                                 # :67-68
    pl2 = torch.sum((points[:, None, :] - ns) * nau, dim=2, keepdim=True)                     # :71
    pl2c = torch.clamp(pl2, torch.zeros_like(pl2), 2 * radius.view(1, -1, 1))                 # :74-75
    pona = ns + pl2c * nau                                                                    # :76
    surf = ppc + rej / safe * radius.view(1, -1, 1)                                           # :79
    fpp = torch.where(perp[..., None], surf, pona)                                            # :82
    dist = torch.norm(points[:, None, :] - fpp, dim=2)                                        # :85
    ci = torch.argmin(dist, dim=1)                                                            # :88
    cd = dist[range(len(points)), ci]
    if move_points_to_mantle:                                                                 # :91-105
        ds = torch.norm(pona - ns, dim=2, keepdim=True)
        de = torch.norm(pona - ne, dim=2, keepdim=True)
        face = torch.where(ds < de, ns, ne)
        fm = torch.where(perp[..., None], surf, face)
        fpp_sel = fm[range(len(points)), ci]
    else:
        fpp_sel = fpp[range(len(points)), ci]
    return IDs[ci].numpy(), cd.numpy(), (fpp_sel - points).numpy()


def generate_offset_cloud(cloud, start, end, radius, IDs, batch_size=1024):
    """Projection.py:117-144 on the CPU."""
    start, end = torch.as_tensor(start, dtype=torch.float32), torch.as_tensor(end, dtype=torch.float32)
    radius, IDs = torch.as_tensor(radius, dtype=torch.float32), torch.as_tensor(IDs, dtype=torch.int32)
    axis = end - start
    axis_length = torch.norm(axis, dim=1, keepdim=True)
    safe = axis_length.clone()
    safe[safe < 1e-8] = 1e-8
    axis_unit = axis / safe
    out = np.zeros((len(cloud), 7))
    for i in range(0, len(cloud), batch_size):
        batch = cloud[i:i + batch_size, :3]
        ids, _, off = closest_cylinder_batch(batch, start, radius, axis_length, axis_unit, IDs)
        out[i:i + batch_size, :3], out[i:i + batch_size, 3:6], out[i:i + batch_size, 6] = batch, off, ids
    return out
