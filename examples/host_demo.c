/*
 * host_demo.c -- a plain C host program against rtk.h, exactly as it would be written for
 * the reference (rtk_build_scene / rtk_trace_ray / rtk_free_scene, reference rtk.h:126-129),
 * plus the additive batch call of rtk_amd.h. Build:
 *
 *   gcc -std=c11 -O2 -Iinclude examples/host_demo.c -Lrtk_amd -lrtk_amd \
 *       -Wl,-rpath,$PWD/rtk_amd -Wl,-rpath,/opt/rocm/lib -lm -o examples/host_demo
 *
 * Prints hit counts of the per-ray and the batch path; exits non-zero if they disagree.
 */
#include <stdio.h>
#include <stdlib.h>

#include "rtk.h"
#include "rtk_amd.h"

static uint64_t splitmix64(uint64_t x)
{
	uint64_t z = x + 0x9E3779B97F4A7C15ull;
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
	return z ^ (z >> 31);
}

static float u01(uint64_t seed, uint64_t k) { return (float)(splitmix64((seed << 40) + k) >> 40) * (1.0f / 16777216.0f); }

int main(int argc, char **argv)
{
	const size_t num_tris = argc > 1 ? (size_t)atol(argv[1]) : 2000;
	const size_t num_rays = argc > 2 ? (size_t)atol(argv[2]) : 4096;

	/* one unindexed float mesh: triangle i uses vertices 3i..3i+2 (rtk.c:1061-1068) */
	float *pos = (float *)malloc(num_tris * 9 * sizeof(float));
	for (size_t i = 0; i < num_tris; i++)
		for (int v = 0; v < 3; v++)
			for (int a = 0; a < 3; a++)
				pos[9 * i + 3 * v + a] = u01(1, 12 * i + a) + 0.05f * (u01(1, 12 * i + 3 + 3 * v + a) - 0.5f);

	rtk_mesh mesh = { 0 };
	mesh.num_triangles = num_tris;
	mesh.position.data = pos;
	mesh.position.type = RTK_TYPE_F32;
	rtk_scene_desc desc = { 0 };
	desc.meshes = &mesh;
	desc.num_meshes = 1;

	rtk_scene *scene = rtk_build_scene(&desc);
	if (!scene) { fprintf(stderr, "rtk_build_scene failed: %s\n", rtk_amd_last_error()); return 2; }
	printf("scene blob: %llu bytes\n", (unsigned long long)scene->size_in_bytes);

	rtk_ray *rays = (rtk_ray *)malloc(num_rays * sizeof(rtk_ray));
	for (size_t i = 0; i < num_rays; i++) {
		rays[i].origin.x = u01(2, 4 * i); rays[i].origin.y = u01(2, 4 * i + 1); rays[i].origin.z = -1.0f;
		rays[i].direction.x = 0.3f * (u01(2, 4 * i + 2) - 0.5f);
		rays[i].direction.y = 0.3f * (u01(2, 4 * i + 3) - 0.5f);
		rays[i].direction.z = 1.0f;
		rays[i].min_t = 0.0f; rays[i].max_t = RTK_INF;
	}

	/* the reference's per-ray call (a GPU batch of one each: correct, slow) on a few rays */
	size_t single_hits = 0, check = num_rays < 64 ? num_rays : 64;
	rtk_hit *one = (rtk_hit *)calloc(check, sizeof(rtk_hit));
	for (size_t i = 0; i < check; i++) single_hits += rtk_trace_ray(scene, &rays[i], &one[i]) ? 1 : 0;

	/* the batch call */
	rtk_hit *hits = (rtk_hit *)calloc(num_rays, sizeof(rtk_hit));
	uint8_t *mask = (uint8_t *)calloc(num_rays, 1);
	const size_t batch_hits = rtk_trace_rays(scene, rays, num_rays, hits, mask);
	if (batch_hits == (size_t)-1) { fprintf(stderr, "rtk_trace_rays failed: %s\n", rtk_amd_last_error()); return 3; }

	int bad = 0;
	size_t batch_prefix_hits = 0;
	for (size_t i = 0; i < check; i++) {
		batch_prefix_hits += mask[i];
		if (mask[i] && (hits[i].triangle_index != one[i].triangle_index || hits[i].t != one[i].t)) bad++;
	}
	printf("rays %zu: batch hits %zu; first %zu rays: per-ray hits %zu, batch hits %zu, mismatches %d\n",
		num_rays, batch_hits, check, single_hits, batch_prefix_hits, bad);
	rtk_free_scene(scene);
	free(pos); free(rays); free(hits); free(mask); free(one);
	return (bad || single_hits != batch_prefix_hits) ? 1 : 0;
}
