"""The BVH built on the GPU (csrc/rt_bvh_build.hip, rt_scene_desc.build_bvh_on_device) against the host-built one.

Which tree the kernels walk must not change a single bit of the result: the closest hit and its tie-break do not
depend on the search order.  So the device-built tree is checked (a) structurally — every slot in exactly one leaf,
every quantised child box containing its subtree, stack depth as declared (the same check_bvh the host builder
passes) — and (b) by rendering: images and first hits bit-identical to the host-built tree's, which the parity
tests pin to the oracle.
"""
import types

import numpy as np
import pytest

import oracle_ffi as O
from test_gpu_parity import primary_and_bounce_rays, quadric_playground
from test_host import check_bvh

pytestmark = pytest.mark.gpu


def same_bits(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))


def device_tree(gpu, flat):
    """What check_bvh wants to see, read back from the device."""
    info = gpu.scene_info()
    nodes, slots = gpu.read_bvh()
    desc = types.SimpleNamespace(n_slots=info["n_slots"], bvh_root_ref=info["bvh_root_ref"], bvh_stack_depth=info["bvh_stack_depth"],
                                 always_begin=flat.desc.always_begin, always_count=flat.desc.always_count)
    return types.SimpleNamespace(nodes=lambda: nodes, slots=lambda: slots, desc=desc), info


PLOC, LBVH = 1, 2     # rt_bvh_build: RT_BVH_DEVICE (clustering), RT_BVH_DEVICE_LBVH (Morton radix tree)


def both_trees(rt, gpu, scene, params, camera, sampler, n_rays, leaf_max=0, expect_boxes=False, algo=PLOC):
    rays = primary_and_bounce_rays(rt, scene, params, camera, n_rays)
    host_flat = scene.flatten()
    gpu.upload_scene(host_flat)
    gpu.upload_sampler(sampler)
    assert gpu.scene_info()["built_on_device"] == 0
    img_host, st_host = gpu.render(params, camera, stats=True)
    hit_host = gpu.trace_first_hit(rays)

    dev_flat = scene.flatten(device_bvh=True)
    assert dev_flat.desc.build_bvh_on_device == 1 and dev_flat.desc.n_bvh_nodes == 0
    dev_flat.desc.bvh_leaf_max = leaf_max
    dev_flat.desc.build_bvh_on_device = algo
    gpu.upload_scene(dev_flat)
    tree, info = device_tree(gpu, dev_flat)
    assert info["built_on_device"] == 1 and info["n_bvh_nodes"] > 0 and info["bvh_build_ms"] > 0.0
    check_bvh(tree, expect_boxes=expect_boxes)
    img_dev, st_dev = gpu.render(params, camera, stats=True)
    hit_dev = gpu.trace_first_hit(rays)
    assert same_bits(img_host, img_dev), "the image depends on which tree was walked"
    for a, b in zip(hit_host, hit_dev):
        assert same_bits(a, b)
    assert st_host["rays"] == st_dev["rays"] and st_host["accepted_hits"] == st_dev["accepted_hits"]
    return info, st_host, st_dev


@pytest.mark.parametrize("algo", [PLOC, LBVH])
@pytest.mark.parametrize("name,width,spp,extra,leaf_max", [
    ("C1", 160, 9, {}, 0),                      # moving spheres: the moving-slot list is translated to the new order
    ("C3", 96, 9, {}, 0),                       # blocks, Translate(RotateY) spheres, media; the rect list
    ("C3", 64, 4, {}, 2),                       # pairs of slots per leaf
    ("C4", 64, 4, {"mesh_grid": 2}, 2),         # 90 584 triangles: attributes are gathered with the slots
])
def test_device_built_tree_renders_the_same_bits(rt, gpu, name, width, spp, extra, leaf_max, algo):
    scene, params, camera, sampler = rt.configs.build(name, image_width=width, samples_per_pixel=spp, **extra)
    info, st_host, st_dev = both_trees(rt, gpu, scene, params, camera, sampler, 1 << 14, leaf_max, expect_boxes=(name == "C1"), algo=algo)
    # Morton splits make a looser tree than binned SAH (more node visits for the same hits); the clustering stays close to it
    assert st_dev["node_visits"] < (4 if algo == LBVH else 2) * st_host["node_visits"]


def test_device_build_next_to_an_always_list(rt, gpu):
    scene, params = quadric_playground(rt)
    sampler = rt.host.MultiJitteredSampler(params.samples_per_pixel)
    for algo in (PLOC, LBVH):
        info, _, _ = both_trees(rt, gpu, scene, params, params.camera(), sampler, 1 << 13, algo=algo)
        assert info["n_slots"] == 31 + 15      # 39 of them in the tree, 8 under a general Transform; 7 quadrics in the always-list


def test_coincident_centres_and_tiny_inputs(rt, gpu):
    """Equal Morton codes are split by position (Karras 2012, sec. 4), equal merge costs fall back to pairing neighbours:
    40 concentric spheres plus 24 identical ones in a row."""
    s = rt.host.Scene()
    m = s.lambertian(s.solid(0.5, 0.5, 0.5))
    objs = [s.sphere((0, 0, 0), 0.1 * (i + 1), s.dielectric(1.5)) for i in range(40)]
    objs += [s.sphere((5 + 0.5 * i, 0, 0), 0.2, m) for i in range(24)]
    s.set_roots(s.hittable_list(objs), s.hittable_list([s.xz_rect(-1, 1, -1, 1, 9, s.diffuse_light(s.solid(0, 0, 0)))]))
    params = rt.host.RaytracerConfig.load().scene_params("CornellBox").replace(
        image_width=64, aspect_ratio=1.0, samples_per_pixel=4, look_from=(3.0, 2.0, 14.0), look_at=(4.0, 0.0, 0.0), vertical_fov=50.0,
        background=(0.5, 0.6, 0.8))
    sampler = rt.host.MultiJitteredSampler(4)
    for leaf_max in (1, 2):
        for algo in (PLOC, LBVH):
            both_trees(rt, gpu, s, params, params.camera(), sampler, 1 << 12, leaf_max, algo=algo)
    # 25 primitives: the smallest scene that gets a tree at all (up to 24 the flattener makes a flat list)
    s2 = rt.host.Scene()
    m2 = s2.lambertian(s2.solid(0.5, 0.5, 0.5))
    objs = [s2.sphere((float(i), 0, 0), 0.4, m2) for i in range(25)]
    s2.set_roots(s2.hittable_list(objs), s2.hittable_list([s2.xz_rect(-1, 1, -1, 1, 9, s2.diffuse_light(s2.solid(0, 0, 0)))]))
    p2 = params.replace(look_from=(12.0, 2.0, 30.0), look_at=(12.0, 0.0, 0.0))
    for algo in (PLOC, LBVH):
        both_trees(rt, gpu, s2, p2, p2.camera(), sampler, 1 << 12, 2, algo=algo)
