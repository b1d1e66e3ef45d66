"""Drop-in shim for the reference's ``NPHM`` package: the neural-field modules resolve to nphm_amd;
every other sub-package (data, evaluation, utils.mesh_operations, env_paths, trainers, losses, ...)
is looked up in any other ``NPHM`` directory on sys.path, i.e. the reference's own ``src/NPHM`` when
it is installed behind this repo."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
