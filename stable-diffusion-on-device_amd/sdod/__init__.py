"""`sdod` -- drop-in Python operator surface of vaenyr/stable-diffusion-on-device
(reference sdod/__init__.py:1-28), backed on MI355X by hand-written gfx950 kernels.

Same public names: EfficientGN, add_module_properties, staticproperty and the lazy module
attributes __version__, __has_repo__, __repo__, __commit__.  The MI355X txt2img pipeline that the
reference runs through QNN graphs lives in the `sdod.amd` subpackage."""
from .efficient_gn import EfficientGN
from .utils import add_module_properties, staticproperty


def _version_attr(name):
    def getter():
        from . import version
        return getattr(version, name)
    return getter


add_module_properties(__name__, {
    '__version__': staticproperty(_version_attr('version')),
    '__has_repo__': staticproperty(_version_attr('has_repo')),
    '__repo__': staticproperty(_version_attr('repo')),
    '__commit__': staticproperty(_version_attr('commit')),
})
