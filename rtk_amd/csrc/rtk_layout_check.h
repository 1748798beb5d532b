// rtk_layout_check.h -- compile-time proof that include/rtk.h has the reference's ABI
// (sizes and offsets probed from the reference build).
#pragma once

#include <stddef.h>

#include "rtk.h"
#include "rtk_amd.h"

static_assert(sizeof(rtk_vec3) == 12, "rtk_vec3");
static_assert(sizeof(rtk_vertex) == 16 && offsetof(rtk_vertex, index) == 12, "rtk_vertex");
static_assert(sizeof(rtk_ray) == 32 && offsetof(rtk_ray, direction) == 12 && offsetof(rtk_ray, min_t) == 24, "rtk_ray");
static_assert(sizeof(rtk_hit) == 68 && offsetof(rtk_hit, vertex) == 12 && offsetof(rtk_hit, mesh_index) == 60, "rtk_hit");
static_assert(sizeof(rtk_buffer) == 24, "rtk_buffer");
static_assert(sizeof(rtk_mesh) == 96 && offsetof(rtk_mesh, position) == 16 && offsetof(rtk_mesh, index) == 40 &&
	offsetof(rtk_mesh, position_cb) == 64 && offsetof(rtk_mesh, index_cb) == 80, "rtk_mesh");
static_assert(sizeof(rtk_scene) == 56 && offsetof(rtk_scene, size_in_bytes) == 24 && offsetof(rtk_scene, vertex_offset) == 48, "rtk_scene");
static_assert(sizeof(rtk_scene_desc) == 32, "rtk_scene_desc");
static_assert(sizeof(rtk_task) == 40 && offsetof(rtk_task, cost) == 16 && offsetof(rtk_task, arg) == 32, "rtk_task");
static_assert(sizeof(rtk_hit_record) == 16, "rtk_hit_record");
