"""Import alias: the package directory is `computer-graphics-engine_amd` (hyphenated,
as the project layout prescribes), which `import` cannot spell. `import cge_amd`
re-exports its modules: cge_amd.abi, cge_amd.capi, cge_amd.scenes, cge_amd.sharding."""
import importlib as _il

_pkg = _il.import_module("computer-graphics-engine_amd")
abi = _il.import_module("computer-graphics-engine_amd.abi")
scenes = _il.import_module("computer-graphics-engine_amd.scenes")
sharding = _il.import_module("computer-graphics-engine_amd.sharding")
capi = _il.import_module("computer-graphics-engine_amd.capi")
Library = capi.Library
