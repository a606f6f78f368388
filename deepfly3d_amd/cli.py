"""`df3d-cli` on the MI355X back-end: same flags, defaults and exit codes as reference df3d/cli.py:15-358.
Video rendering flags are accepted and reported as unsupported (visualisation is out of scope)."""
import argparse
import logging
import sys
from collections import deque
from pathlib import Path

from . import logger
from .core import Core


def parse_cli_args(argv=None):
    p = argparse.ArgumentParser(description="DeepFly3D pose estimation (MI355X back-end)")
    p.add_argument("-v", "--verbose", help="Enable info output (such as progress bars)", action="store_true")
    p.add_argument("-vv", "--verbose2", help="Enable debug output", action="store_true")
    p.add_argument("-d", "--debug", help="Displays the argument list for debugging purposes", action="store_true")
    p.add_argument("input_folder", help="Without additional arguments, a folder containing unlabeled images.", metavar="INPUT")
    p.add_argument("--output-folder", default=None,
                   help="Folder where results are written; default: INPUT suffixed with '_df3d'.")
    p.add_argument("-r", "--recursive", help="INPUT is a folder. Successively use its subfolders named 'images/'", action="store_true")
    p.add_argument("-f", "--from-file", help="INPUT is a text-file, where each line names a folder.", action="store_true")
    p.add_argument("-x", "--delete-images", help="Delete expanded image files after running (only if the .mp4 exists).", action="store_true")
    p.add_argument("-n", "--num-images-max", help="Maximal number of images to process (0 = all).", default=0, type=int)
    p.add_argument("--order", "--camera-ids", help="Ordering of the cameras, e.g. --order 0 1 4 3 2 5 6.",
                   default=[0, 1, 2, 3, 4, 5, 6], type=int, nargs="*")
    p.add_argument("--video-2d", help="Generate pose2d videos", action="store_true")
    p.add_argument("--video-3d", help="Generate pose3d videos", action="store_true")
    p.add_argument("--skip-pose-estimation", help="Skip 2D and 3D pose estimation", dest="skip_estimation", action="store_true")
    p.add_argument("--batch-size", help="Batch size for inference", type=int, default=8)
    p.add_argument("--pin-memory-disabled", help="Disable pinned host staging buffers", action="store_true")
    p.add_argument("--output-fps", help="FPS for output videos.", type=float, default=None)
    p.add_argument("--dtype", choices=["f32", "f32s", "f16", "bf16"], default="f32",
                   help="hourglass arithmetic on the GPU: f32 (default: the reference's arithmetic); f32s = float32 tensors, weights and accumulation with "
                        "every product formed from IEEE-half splits on the 16-bit matrix cores, ~2.2x faster, heat-maps ~2e-6 of their range from f32's "
                        "(MEASURED on seeded synthetic weights; operands must lie inside the half range 65 504, which batch-normalised activations do); f16 = IEEE-half activations and weights on the "
                        "matrix cores with fp32 accumulation, ~6x faster; MEASURED on seeded synthetic weights against the fp32 oracle: heat-map confidences "
                        "6-8e-4 off on peaked maps, up to 2.8e-3 on flat ones (the reference's test tolerance is 2e-3, tests/test_df3d.py:173-178; "
                        "with the trained checkpoint unverified: tests/test_gpu_reference_pin.py decides once weights are present); bf16 = same speed, "
                        "fp32's exponent range, 8 significant bits: confidences ~6e-3 off on peaked maps, outside that tolerance.  f16 / f32s refuse weights beyond "
                        "the half range when they are loaded, and if an activation overflows on real images (an infinity or a NaN in any heat-map) the run "
                        "stops with an error that says so instead of writing df3d_result.pkl: rerun with --dtype f32")
    args = p.parse_args(argv)
    inp = Path(args.input_folder).expanduser().resolve()
    args.output_folder = str(inp.with_name(inp.stem + "_df3d")) if args.output_folder is None else str(Path(args.output_folder).expanduser().resolve())
    args.input_folder = str(inp)
    return args


def setup_logger(args):
    handler = logging.StreamHandler()
    handler.setLevel(logging.DEBUG)
    lg = logger.getLogger()
    lg.addHandler(handler)
    lg.setLevel(logging.DEBUG if args.verbose2 else logging.INFO if args.verbose else logging.WARNING)


def print_debug(args):
    print(f"Enabled logging level: {logging.getLevelName(logger.getLogger().getEffectiveLevel())}")
    print("Arguments are:")
    for k, v in vars(args).items():
        print(f"\t{k}: {v}")
    print()
    return 0


def run(args):
    if args.skip_estimation and not args.video_2d and not args.video_3d:
        logger.info("Nothing to do. Check your command-line arguments.")
        return 0
    logger.info(f"\nWorking in {args.input_folder}")
    core = Core(args.input_folder, args.output_folder, args.num_images_max, args.order, dtype=args.dtype, device=getattr(args, "device", None))
    if not args.skip_estimation:
        core.pose2d_estimation(args.batch_size, args.pin_memory_disabled)
        core.save()
    core.calibrate_calc(0, core.max_img_id)
    core.save()
    if args.video_2d or args.video_3d:
        # f4 (reference cli.py:305-321): frames drawn on the GPU (csrc/render.hip), encoded by ffmpeg when present.  Rank 0 draws and
        # encodes; the peers wait for its outcome with a heartbeat (distributed.primary_section), so that an encoder failure moves
        # every rank on to the next folder together and a long encode never holds a peer in one collective
        from . import distributed as dd
        from . import video

        fps = args.output_fps if args.output_fps is not None else core.fps

        def videos(beat):
            if args.video_2d:
                video.make_pose2d_video(core, fps=fps, progress=beat)
            if args.video_3d:
                video.make_pose3d_video(core, fps=fps, progress=beat)

        dd.primary_section(videos, "video")
    if args.delete_images:
        core.delete_images()
    return 0


def run_in_folders(args, folders):
    errors = []
    for folder in folders:
        try:
            args.input_folder = str(folder)  # like the reference, every folder writes into the one output folder
            run(args)
        except KeyboardInterrupt:
            logger.warning("Keyboard Interrupt received. Terminating...")
            break
        except Exception as e:  # per-folder isolation, as the reference does
            errors.append((folder, e))
            logger.error(f"An error occured while processing {folder}. Continuing...")
    if errors:
        logger.error(f"\n{len(errors)} out of {len(folders)} folders terminated with errors.")
        for folder, exc in errors:
            logger.error(f"\nIn {folder}", exc_info=exc)


def find_subfolders(path, name):
    """Breadth-first search for sub-folders called `name`, not descending into matches."""
    found, queue, seen = [], deque([Path(path)]), set()
    while queue:
        cur = queue.popleft()
        if cur.is_dir() and cur not in seen:
            seen.add(cur)
            if cur.name == name:
                found.append(str(cur))
            else:
                queue.extend(cur.iterdir())
    return found


def run_recursive(args):
    sub = find_subfolders(args.input_folder, "images")
    logger.info(f"Found {len(sub)} subfolder(s):\n-" + "\n-".join(sub))
    args.recursive = False
    run_in_folders(args, sub)


def run_from_file(args):
    try:
        with open(args.input_folder, "r") as f:
            folders = [line.strip() for line in f]
    except FileNotFoundError:
        logger.error(f"Unable to find the file {args.input_folder}")
        return 1
    except IsADirectoryError:
        logger.error(f"{args.input_folder} is a directory, please provide a file instead.")
        return 1
    folders = [Path(f) for f in dict.fromkeys(folders) if f.strip()]
    bad = [f for f in folders if not f.is_dir()]
    for f in bad:
        logger.error(f"[Error] Not a directory or does not exist: {f}")
    if bad:
        return 1
    args.from_file = False
    run_in_folders(args, folders)


def main(argv=None):
    """Entry point.  Multi-GPU: `python -m torch.distributed.run --nproc-per-node N -m deepfly3d_amd.cli INPUT ...`
    (one process per GPU; frames are sharded, rank 0 writes the result)."""
    import os

    args = parse_cli_args(argv)
    setup_logger(args)
    if args.debug:
        return print_debug(args)
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        import torch.distributed as dist

        from . import distributed as dd

        _, _, local_rank = dd.init_from_env()
        args.device = str(dd.local_device(local_rank))
        try:
            return _dispatch(args)
        finally:
            dist.barrier()
            dist.destroy_process_group()
    return _dispatch(args)


def _dispatch(args):
    if args.from_file and args.recursive:
        logger.error('Error: choose an input method between "from file" and "recursive" but not both.')
        return 1
    if args.recursive:
        return run_recursive(args)
    if args.from_file:
        return run_from_file(args)
    return run(args)


if __name__ == "__main__":
    sys.exit(main())
