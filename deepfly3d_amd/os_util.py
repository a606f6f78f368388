"""Image-folder discovery (drop-in behaviour of reference df3d/os_util.py:7-59)."""
import os
import re

NUM_CAMERAS = 7
_UPPER = 100000  # the reference's search bound for image ids


def image_name(cam_id, img_id, pad=False):
    return f"camera_{cam_id}_img_{img_id:06d}" if pad else f"camera_{cam_id}_img_{img_id}"


def image_path_for(folder, cam_id, img_id):
    """Path of an existing frame, trying the unpadded then the 6-digit padded name."""
    for pad in (False, True):
        p = os.path.join(folder, image_name(cam_id, img_id, pad) + ".jpg")
        if os.path.isfile(p):
            return p
    raise FileNotFoundError(f"no image for camera {cam_id}, frame {img_id} in {folder}")


def frame_exists(folder, img_id):
    """True if any camera has this frame (unpadded name) or camera 0 has it (padded name)."""
    if any(os.path.isfile(os.path.join(folder, image_name(c, img_id) + ".jpg")) for c in range(NUM_CAMERAS)):
        return True
    return os.path.isfile(os.path.join(folder, image_name(0, img_id, True) + ".jpg"))


def get_max_img_id(folder):
    """Largest frame id present, by bisection over [0, 100000) (frames are assumed contiguous from 0)."""
    lo, hi = 0, _UPPER
    while hi - lo > 1:
        mid = (lo + hi) // 2
        if frame_exists(folder, mid):
            lo = mid
        else:
            hi = mid
    if not frame_exists(folder, lo):
        raise FileNotFoundError("No image found.")
    return lo


def parse_img_name(name):
    m = re.match(r"camera_(\d+)_img_(\d+)", name.replace(".jpg", ""))
    return int(m[1]), int(m[2])


def parse_vid_name(name):
    m = re.match(r"camera_(\d+)", name.replace(".mp4", ""))
    return int(m[1])
