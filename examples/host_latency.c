/*
 * host_latency.c -- what the reference's per-ray call costs a C host: rtk_trace_ray in a loop, from one thread and from
 * several at once (the reference call is re-entrant, rtk.c:543-577; so is this one). Build like host_demo.c, plus -lpthread:
 *
 *   gcc -std=c11 -O2 -Iinclude examples/host_latency.c -Lrtk_amd -lrtk_amd -lpthread \
 *       -Wl,-rpath,$PWD/rtk_amd -Wl,-rpath,/opt/rocm/lib -lm -o examples/host_latency
 *
 * Usage: host_latency [triangles] [calls per thread] [threads]
 */
#define _POSIX_C_SOURCE 200809L
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

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
static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }

struct job { const rtk_scene *scene; size_t calls; unsigned id; size_t hits; };

static void *worker(void *arg)
{
	struct job *j = (struct job *)arg;
	for (size_t i = 0; i < j->calls; i++) {
		rtk_ray r;
		rtk_hit h;
		r.origin.x = u01(7 + j->id, 4 * i); r.origin.y = u01(7 + j->id, 4 * i + 1); r.origin.z = -1.0f;
		r.direction.x = 0.3f * (u01(7 + j->id, 4 * i + 2) - 0.5f);
		r.direction.y = 0.3f * (u01(7 + j->id, 4 * i + 3) - 0.5f);
		r.direction.z = 1.0f;
		r.min_t = 0.0f; r.max_t = RTK_INF;
		if (rtk_trace_ray(j->scene, &r, &h)) j->hits++;
	}
	return NULL;
}

int main(int argc, char **argv)
{
	const size_t num_tris = argc > 1 ? (size_t)atol(argv[1]) : 100000;
	const size_t calls = argc > 2 ? (size_t)atol(argv[2]) : 2000;
	const unsigned max_threads = argc > 3 ? (unsigned)atoi(argv[3]) : 8;
	float *pos = (float *)malloc(num_tris * 9 * sizeof(float));
	for (size_t i = 0; i < num_tris; i++)
		for (int v = 0; v < 3; v++)
			for (int a = 0; a < 3; a++)
				pos[9 * i + 3 * v + a] = u01(1, 12 * i + a) + 0.02f * (u01(1, 12 * i + 3 + 3 * v + a) - 0.5f);
	rtk_mesh mesh = { 0 };
	mesh.num_triangles = num_tris;
	mesh.position.data = pos;
	mesh.position.type = RTK_TYPE_F32;
	rtk_scene_desc desc = { 0 };
	desc.meshes = &mesh;
	desc.num_meshes = 1;
	rtk_scene *scene = rtk_build_scene(&desc);
	if (!scene) { fprintf(stderr, "rtk_build_scene failed: %s\n", rtk_amd_last_error()); return 2; }
	{ struct job warm = { scene, 200, 99, 0 }; worker(&warm); }          /* first call uploads / warms up */
	for (unsigned nt = 1; nt <= max_threads; nt *= 2) {
		pthread_t th[64];
		struct job jobs[64];
		const double t0 = now_s();
		for (unsigned k = 0; k < nt; k++) { jobs[k].scene = scene; jobs[k].calls = calls; jobs[k].id = k; jobs[k].hits = 0; pthread_create(&th[k], NULL, worker, &jobs[k]); }
		size_t hits = 0;
		for (unsigned k = 0; k < nt; k++) { pthread_join(th[k], NULL); hits += jobs[k].hits; }
		const double dt = now_s() - t0;
		printf("rtk_trace_ray: %u thread(s) x %zu calls: %.1f us per call per thread, %.1f us aggregate (%.0f rays/s), %zu hits\n",
			nt, calls, 1e6 * dt / (double)calls, 1e6 * dt / (double)(calls * nt), (double)(calls * nt) / dt, hits);
	}
	rtk_free_scene(scene);
	free(pos);
	return 0;
}
