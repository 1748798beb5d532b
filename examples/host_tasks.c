/*
 * host_tasks.c -- a C host that drives the reference's caller-scheduled build (rtk_start_build / rtk_run_task,
 * reference rtk.h:119-120, rtk.c:1692-1717) from its OWN thread pool, then traces the result on the GPU(s):
 *
 *   1. rtk_amd_set_builder(RTK_AMD_BUILDER_CPU_TASKS): the build is a real task graph (setup ranges, one SAH task
 *      per node, vertex-group finalisation); four pthreads pull tasks from a shared list until none is pending.
 *      With the default device builder the same loop runs exactly one task.
 *   2. rtk_finish_build -> a scene blob in the reference's format; rtk_trace_rays traces it on the GPU.
 *   3. the same rays through the single-process multi-GPU context (rtk_mgpu_*), scene replicated, rays sharded.
 *   4. rtk_trace_rays_filter with a host callback that rejects every other triangle.
 *
 *   gcc -std=c11 -O2 -Iinclude examples/host_tasks.c -Lrtk_amd -lrtk_amd -lpthread \
 *       -Wl,-rpath,$PWD/rtk_amd -Wl,-rpath,/opt/rocm/lib -lm -o examples/host_tasks
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

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

/* -- the host's scheduler: a locked stack of tasks, workers run until nothing is queued and nobody is running -- */
#define MAX_TASKS 65536
static rtk_task g_tasks[MAX_TASKS];
static size_t g_num_tasks, g_running, g_tasks_run;
static pthread_mutex_t g_lock = PTHREAD_MUTEX_INITIALIZER;

static void *worker(void *arg)
{
	rtk_task spawned[256];
	(void)arg;
	for (;;) {
		rtk_task t;
		int have = 0;
		pthread_mutex_lock(&g_lock);
		if (g_num_tasks) { t = g_tasks[--g_num_tasks]; g_running++; have = 1; }
		else if (g_running == 0) { pthread_mutex_unlock(&g_lock); return NULL; }
		pthread_mutex_unlock(&g_lock);
		if (!have) continue;
		const size_t n = rtk_run_task(&t, spawned, 256);
		pthread_mutex_lock(&g_lock);
		memcpy(g_tasks + g_num_tasks, spawned, n * sizeof(rtk_task));
		g_num_tasks += n;
		g_running--;
		g_tasks_run++;
		pthread_mutex_unlock(&g_lock);
	}
}

static bool every_other_triangle(void *user, const rtk_ray *ray, const rtk_hit *hit)
{
	(void)user; (void)ray;
	return (hit->triangle_index & 1u) == 0u;
}

int main(int argc, char **argv)
{
	const size_t num_tris = argc > 1 ? (size_t)atol(argv[1]) : 100000;
	const size_t num_rays = argc > 2 ? (size_t)atol(argv[2]) : 65536;
	float *pos = (float *)malloc(num_tris * 9 * sizeof(float));
	for (size_t i = 0; i < num_tris; i++)
		for (int v = 0; v < 3; v++)
			for (int a = 0; a < 3; a++)
				pos[9 * i + 3 * v + a] = u01(1, 12 * i + a) + 0.03f * (u01(1, 12 * i + 3 + 3 * v + a) - 0.5f);
	rtk_mesh mesh = { 0 };
	mesh.num_triangles = num_tris;
	mesh.position.data = pos;
	mesh.position.type = RTK_TYPE_F32;
	rtk_scene_desc desc = { 0 };
	desc.meshes = &mesh;
	desc.num_meshes = 1;

	/* 1. the task graph on the host's threads */
	if (rtk_amd_set_builder(RTK_AMD_BUILDER_CPU_TASKS) != RTK_AMD_OK) return 2;
	rtk_build *build = rtk_start_build(&desc, &g_tasks[0]);
	if (!build) { fprintf(stderr, "rtk_start_build: %s\n", rtk_amd_last_error()); return 2; }
	g_num_tasks = 1;
	pthread_t th[4];
	for (int i = 0; i < 4; i++) pthread_create(&th[i], NULL, worker, NULL);
	for (int i = 0; i < 4; i++) pthread_join(th[i], NULL);
	rtk_scene *scene = rtk_finish_build(build);
	if (!scene) { fprintf(stderr, "rtk_finish_build: %s\n", rtk_amd_last_error()); return 2; }
	printf("task graph: %zu tasks run on 4 threads, blob %llu bytes\n", g_tasks_run, (unsigned long long)scene->size_in_bytes);

	rtk_ray *rays = (rtk_ray *)malloc(num_rays * sizeof(rtk_ray));
	for (size_t i = 0; i < num_rays; i++) {
		rays[i].origin.x = u01(2, 4 * i); rays[i].origin.y = u01(2, 4 * i + 1); rays[i].origin.z = -1.0f;
		rays[i].direction.x = 0.3f * (u01(2, 4 * i + 2) - 0.5f);
		rays[i].direction.y = 0.3f * (u01(2, 4 * i + 3) - 0.5f);
		rays[i].direction.z = 1.0f;
		rays[i].min_t = 0.0f; rays[i].max_t = RTK_INF;
	}

	/* 2. trace the CPU-built blob on the GPU */
	rtk_hit *hits = (rtk_hit *)calloc(num_rays, sizeof(rtk_hit));
	uint8_t *mask = (uint8_t *)calloc(num_rays, 1);
	const size_t nhit = rtk_trace_rays(scene, rays, num_rays, hits, mask);
	if (nhit == (size_t)-1) { fprintf(stderr, "rtk_trace_rays: %s\n", rtk_amd_last_error()); return 3; }

	/* 3. the same batch through the multi-GPU context: scene replicated by the DEVICE builder on every GPU, rays sharded */
	rtk_amd_set_builder(RTK_AMD_BUILDER_DEVICE);
	int bad = 0, ties = 0;
	rtk_mgpu *ctx = rtk_mgpu_create(NULL, 0);
	if (!ctx || rtk_mgpu_build(ctx, &desc) != RTK_AMD_OK) { fprintf(stderr, "rtk_mgpu: %s\n", rtk_amd_last_error()); return 4; }
	rtk_hit_record *rec = (rtk_hit_record *)calloc(num_rays, sizeof(rtk_hit_record));
	if (rtk_mgpu_trace_rays(ctx, rays, num_rays, rec, NULL) != RTK_AMD_OK) { fprintf(stderr, "rtk_mgpu_trace_rays: %s\n", rtk_amd_last_error()); return 4; }
	for (size_t i = 0; i < num_rays; i++) {
		const int hit = rec[i].prim != RTK_PRIM_NONE;
		/* two different BVHs of the same triangles: same triangle, t within 1e-5 (a triangle's t can move by an ulp with
		 * the leaf it sits in: rtk.c's group-of-four double-precision rule, rtk.c:302-336) */
		const float dt = hit ? rec[i].t - hits[i].t : 0.0f;
		if (hit != mask[i] || (hit && (dt < 0 ? -dt : dt) > 1e-5f * hits[i].t)) bad++;
		else if (hit && rec[i].prim != hits[i].triangle_index) ties++;   /* two triangles within an ulp: either is the reference's answer for SOME leaf grouping */
	}
	printf("%d GPU(s): %zu rays, %zu hits, %d differences between the CPU-built and the device-built scene (%d near ties)\n",
		rtk_mgpu_num_devices(ctx), num_rays, nhit, bad, ties);
	rtk_mgpu_destroy(ctx);

	/* 4. host-callback filter: closest hit on an even-numbered triangle */
	rtk_hit *fh = (rtk_hit *)calloc(num_rays, sizeof(rtk_hit));
	uint8_t *fm = (uint8_t *)calloc(num_rays, 1);
	const size_t fhit = rtk_trace_rays_filter(scene, rays, num_rays, fh, fm, every_other_triangle, NULL);
	if (fhit == (size_t)-1) { fprintf(stderr, "rtk_trace_rays_filter: %s\n", rtk_amd_last_error()); return 5; }
	for (size_t i = 0; i < num_rays; i++) {
		if (fm[i] && ((fh[i].triangle_index & 1u) || (mask[i] && fh[i].t < hits[i].t))) bad++;
		if (fm[i] && !mask[i]) bad++;
	}
	printf("filter: %zu of %zu hits survive\n", fhit, nhit);
	rtk_free_scene(scene);
	free(pos); free(rays); free(hits); free(mask); free(rec); free(fh); free(fm);
	return bad ? 1 : 0;
}
