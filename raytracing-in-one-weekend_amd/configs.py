"""The BASELINE.json workloads as (scene, RaytracerParams) pairs.

Cameras come from data/config/raytracer.config.ron where the reference ships one, with the
image size / spp of BASELINE.json on top:
  C1  RandomWorld    1200x675   10 spp   camera of stats.txt:1 (vfov 20, aspect 16/9)
  C2  CornellBox      600x600 1000 spp   default_params camera, aspect 1.0
  C3  Chapter2Final   800x800  10k spp   config camera
  C4  MeshTest       1920x1080 256 spp   config camera, K x K teapots (K=10 -> 2.26 M triangles)
  C5  Chapter2Final  3840x2160 16384 spp (= 128^2, "16k")  config camera at 16:9; the 8-GPU workload: rank g of 8 renders
                                          samples [2048 g, 2048 (g + 1)) of the whole frame (sharding.sample_range)
and the reference's two remaining scenes with their shipped configs:
  GP      GeometricPrimitives  800x600 256 spp  SphericalPanoramic camera; blocks, cylinders, cones, paraboloids
  BLOCKS  Blocks               800x600 256 spp  three checkered blocks under general Transforms
  TWO / PERLIN / LIGHT / EARTH  TwoSpheres, PerlinSpheres, SimpleLight, TexturedSpheres (main.rs:201-360): a handful of spheres each
"""
from . import host

SEED = 0x5EED

WORKLOADS = {
    "C1": dict(scene="RandomWorld", image_width=1200, aspect_ratio=1200.0 / 675.0, samples_per_pixel=10, vertical_fov=20.0),
    "C2": dict(scene="CornellBox", image_width=600, aspect_ratio=1.0, samples_per_pixel=1000),
    "C3": dict(scene="Chapter2Final", image_width=800, aspect_ratio=1.0, samples_per_pixel=10000),
    "C4": dict(scene="MeshTest", image_width=1920, aspect_ratio=1920.0 / 1080.0, samples_per_pixel=256, mesh_grid=10),
    "C5": dict(scene="Chapter2Final", image_width=3840, aspect_ratio=3840.0 / 2160.0, samples_per_pixel=16384),
    "GP": dict(scene="GeometricPrimitives"),
    "BLOCKS": dict(scene="Blocks"),
    "TWO": dict(scene="TwoSpheres"),
    "PERLIN": dict(scene="PerlinSpheres"),
    "LIGHT": dict(scene="SimpleLight"),
    # not in the reference's defined_scenes (and not reachable from its scene switch): default_params, camera pulled back
    # scene_cornell_box_smoke (main.rs:524-683): defined in the reference, reached by no `Scene` variant; Cornell's camera
    "SMOKE": dict(scene="CornellBoxSmoke", config_scene="CornellBox", image_width=600, aspect_ratio=1.0),
    "EARTH": dict(scene="TexturedSpheres", config_scene="CornellBox", look_from=(13.0, 2.0, 3.0), look_at=(0.0, 0.0, 0.0),
                  vertical_fov=20.0, background=(0.7, 0.8, 1.0)),
}


def workload(name, config=None, **overrides):
    """Returns (scene_name, RaytracerParams, mesh_grid)."""
    w = dict(WORKLOADS[name])
    w.update(overrides)
    scene = w.pop("scene")
    mesh_grid = w.pop("mesh_grid", 1)
    config_scene = w.pop("config_scene", scene)
    cfg = config or host.RaytracerConfig.load()
    params = cfg.scene_params(config_scene).replace(**w)
    return scene, params, mesh_grid


def build(name, config=None, seed=SEED, **overrides):
    """Build everything a render needs: (Scene, RaytracerParams, rt_camera, MultiJitteredSampler)."""
    scene_name, params, mesh_grid = workload(name, config, **overrides)
    need = {"RandomWorld": (), "CornellBox": (), "CornellBoxSmoke": (), "Chapter2Final": ("earthmap",),
            "MeshTest": ("uvgrid01", "uvgrid09", "teapot"), "Blocks": ("uvgrid01",),
            "GeometricPrimitives": ("uvgrid01", "uvgrid03", "uvgrid08", "uvgrid10"), "TwoSpheres": (), "SimpleLight": (),
            "PerlinSpheres": ("uvgrid01", "uvgrid03"), "TexturedSpheres": ("earthmap",)}[scene_name]
    assets = host.Assets(mesh_grid=mesh_grid, need=need) if need else None
    scene = host.Scene.build(scene_name, seed, assets)
    sampler = host.MultiJitteredSampler(params.samples_per_pixel, seed=seed)
    return scene, params, params.camera(), sampler
