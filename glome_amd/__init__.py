"""glome_amd -- MI355X-native ray-tracing core behind glome's scene vocabulary.

`glome_amd.api` binds the C ABI (include/glome_hip.h); `glome_amd.scene` records a scene once in glome's
constructor vocabulary so the same description can be replayed into any backend that offers those constructors;
`glome_amd.scenes` holds the benchmark scenes S1..S5.  No compute happens in Python and
there is no CPU fallback: without the built HIP library the API raises.
"""
from .api import (Builder, Context, GlomeError, Scene, camera, camera_from_vectors, compose, deg, light,  # noqa: F401
                  render_params, rotate, scale, translate, xyz_to_uvw)
