"""Host glue the reference drivers import next to the model plugin (reference tools/utils.py:4-34)."""
import os
import re


def get_latest_checkpoint(checkpoint_dir):
    """(path, epoch) of the ``*_<epoch>.pth`` file with the largest epoch suffix."""
    best = None
    for name in os.listdir(checkpoint_dir):
        if not name.endswith(".pth"):
            continue
        m = re.search(r"_(\d+)\.pth$", name)
        epoch = int(m.group(1)) if m else -1
        if best is None or epoch > best[1]:
            best = (os.path.join(checkpoint_dir, name), epoch)
    if best is None:
        raise FileNotFoundError(f"No checkpoint files found in directory: {checkpoint_dir}")
    return best


resolutions = {"350": (350, 630), "360": (360, 640), "720": (720, 1280), "1080": (1080, 1920),
               "1440": (1440, 2560), "2k": (1440, 2560), "2160": (2160, 3840), "4k": (2160, 3840)}
