"""TEST INFRASTRUCTURE ONLY — CPU restatement (pure torch, differentiable) of the reference's
GaussianModel.get_matchloss_from_renderdepth (scene/gaussian_model.py:241-282) for ONE view pair.  The reference
method itself cannot be imported here (scene/gaussian_model.py needs plyfile, cv2, pytorch3d at import time), so
this follows it line by line: PARITY UNPINNED (restatement), every step cites the line it mirrors."""
import torch
import torch.nn.functional as F


def match_loss_pair(depth0, uv0, rays_o, rays_d, cam_rays_d, mask0, mask1, intr1, w2c1, uv1, width, height):
    """depth0 (H,W); returns the pair's contribution to match_loss (:280)."""
    valid_mask = (mask0 * mask1) > 0                                                    # :249-251
    norm_x = (uv0[:, 0] / width) * 2 - 1                                                # :254
    norm_y = (uv0[:, 1] / height) * 2 - 1                                               # :255
    grid = torch.stack([norm_x, norm_y], dim=-1).unsqueeze(0).unsqueeze(0)              # :256
    md = F.grid_sample(depth0.unsqueeze(0).unsqueeze(0), grid, mode="bilinear", align_corners=False).reshape(-1)  # :257-258
    zval0 = (md / cam_rays_d[:, 2]).unsqueeze(-1)                                       # :262
    world = (rays_o + rays_d * zval0).permute(1, 0)                                     # :263
    cam = torch.matmul(w2c1, torch.cat([world, torch.ones_like(world[:1])]))[:3]       # :267
    xyz = torch.matmul(intr1, cam)                                                      # :268
    xy = xyz[:2] / (xyz[2:] + 1e-8)                                                     # :270
    m = (xy[0] > 0) & (xy[0] < width) & (xy[1] > 0) & (xy[1] < height)                  # :271
    xy1 = uv1.permute(1, 0)                                                             # :273
    cur = ((xy - xy1).abs() / torch.tensor([width, height]).type_as(xy1).reshape(2, 1)).mean(dim=0)   # :277
    w = m.float() * valid_mask.float()
    return (cur * w).sum() / (w.sum() + 1e-8)                                           # :279
