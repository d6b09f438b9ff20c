"""Seeded inputs shared by make_golden.py (generator) and the tests (consumers)."""
import torch


def tta_new_object_label(out_hw):
    """Label map [1,1,H0,W0] introducing object id 4 (a rectangle) mid-clip."""
    lab = torch.zeros(1, 1, *out_hw)
    lab[:, :, out_hw[0] // 8: out_hw[0] // 3, out_hw[1] // 2: out_hw[1] * 3 // 4] = 4
    return lab


def ignore_region_label(label0):
    """Reference-frame label with an ignore (255) rectangle that overlaps object 1 and the
    background: exercises the rule that add_reference_frame passes NO ignore mask to
    assign_identity (aot_engine.py:304 -> :209-213), so 255 pixels carry no channel at all."""
    lab = label0.clone()
    H, W = lab.shape[-2:]
    lab[:, :, H // 8: H // 3, W // 5: W // 2] = 255
    return lab


def tta_scaled_images(imgs, hw):
    """The clip's frames at another network size (the multi-scale copies of test-time augmentation): bicubic, computed on
    the CPU so that the generator and the tests feed identical bytes.  (The reference's dataloader resizes the decoded
    image with cv2.INTER_CUBIC, dataloaders/video_transforms.py:625-641; which cubic kernel made the input is immaterial
    to the path under test.)"""
    import torch.nn.functional as F
    return [F.interpolate(im.cpu().float(), size=tuple(hw), mode="bicubic", align_corners=False) for im in imgs]
