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


# ---- camera_<id>.mp4 companions of an image folder (reference df3d/core.py:405-475) -------------------------------
def camera_videos(folder, any_id=True):
    """Sorted `camera_<c>.mp4` files of `folder` (`any_id=False`: single-digit ids only, the reference's delete rule)."""
    import glob

    return sorted(glob.glob(os.path.join(folder, "camera_?.mp4" if any_id else "camera_[0-9].mp4")))


def probe_frame_rate(video):
    """Average frame rate string ffprobe reports for the first video stream, or None when ffprobe is unusable."""
    import subprocess

    cmd = ["ffprobe", "-v", "error", "-select_streams", "v:0", "-show_entries", "stream=avg_frame_rate",
           "-of", "default=noprint_wrappers=1:nokey=1", video]
    try:
        return cmd, subprocess.check_output(cmd, text=True)
    except Exception:
        return cmd, None


def parse_frame_rate(text):
    """'30' -> 30.0, '30000/1001' -> 29.97, '0/0' or garbage -> None."""
    text = text.strip()
    for convert in (float, lambda t: (lambda n, d: n / d if d else None)(*map(int, t.split("/")))):
        try:
            return convert(text)
        except (ValueError, TypeError):
            continue
    return None


def extract_frames(video, folder, cam_id):
    """ffmpeg: camera_<c>.mp4 -> camera_<c>_img_<n>.jpg, numbered from 0, quality scale 2."""
    import subprocess

    pattern = os.path.join(folder, f"camera_{cam_id}_img_%d.jpg")
    return subprocess.call(f"ffmpeg -nostats -loglevel error -i {video} -qscale:v 2 -start_number 0 {pattern}  < /dev/null", shell=True)
