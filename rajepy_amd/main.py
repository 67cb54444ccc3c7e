"""Command line of the radiative-transfer core, same flags as the reference's main.py:16-37:

    python -m rajepy_amd.main [-v] [-rt] [-so] [-r] [-c] model_params.py pipeline_params.py

Several GPUs of one node: launch one process per GPU and the epochs of the run table are
shared out between them (each rank writes the products of its runs, rank 0 the state files):

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        -m rajepy_amd.main -rt model_params.py pipeline_params.py
"""
import argparse
import os
import runpy
import shutil
import time

from . import logger
from .classes import JetModel, Pipeline


def main(argv=None):
    parser = argparse.ArgumentParser(prog="rajepy_amd")
    parser.add_argument("model_param_file", help="Full path to model parameter file")
    parser.add_argument("pipeline_param_file", help="Full path to pipeline parameter file")
    parser.add_argument("-v", "--verbose", action="store_true",
                        help="Increase output verbosity")
    parser.add_argument("-rt", "--radiative-transfer", action="store_true",
                        help="Compute radiative transfer solutions")
    parser.add_argument("-so", "--simobserve", action="store_true",
                        help="Conduct synthetic observations using CASA (not part of this core)")
    parser.add_argument("-r", "--resume", action="store_true",
                        help="Resume previous pipeline run if present")
    parser.add_argument("-c", "--clobber", action="store_true",
                        help="Overwrite any data products/files present")
    parser.add_argument("--storage", choices=("f64", "f32"), default="f64",
                        help="HBM storage width of the 3-D fields")
    args = parser.parse_args(argv)
    jet_file = os.path.abspath(args.model_param_file)
    pline_file = os.path.abspath(args.pipeline_param_file)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1:
        import torch
        import torch.distributed as dist
        local = int(os.environ.get("LOCAL_RANK", "0"))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RJP_DIST_BACKEND=gloo: rehearsal (e.g. several ranks sharing one GPU)
        backend = os.environ.get("RJP_DIST_BACKEND", "nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    dcy = os.path.expanduser(runpy.run_path(pline_file)["params"]['dcys']['model_dcy'])
    os.makedirs(dcy, exist_ok=True)
    log = logger.Log(os.sep.join([dcy, "ModelRun_" + time.strftime(
        "%Y-%m-%d-%H:%M:%S", time.localtime()) + ("_rank%d" % rank if world > 1 else "") +
        ".log"]), verbose=args.verbose)
    pline = Pipeline(JetModel(jet_file, log=log, storage=args.storage), pline_file, log=log)
    pline.log.add_entry("INFO", "Pipeline initiated using model parameters defined in {}, and "
                                "pipeline parameters defined in {}".format(jet_file, pline_file))
    pline.execute(resume=args.resume, clobber=args.clobber, simobserve=args.simobserve,
                  verbose=args.verbose, dryrun=not args.radiative_transfer)
    for f in (jet_file, pline_file) if rank == 0 else ():
        dest = os.path.expanduser(os.sep.join([pline.params['dcys']['model_dcy'],
                                               os.path.basename(f)]))
        if f != dest:
            try:
                shutil.copyfile(f, dest)
            except shutil.SameFileError:
                print(f"{f} and {dest} are the same file")
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()
    return pline


if __name__ == '__main__':
    main()
