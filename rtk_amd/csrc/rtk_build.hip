// rtk_build.hip -- scene construction entry points (rtk.h:119-127). PLACEHOLDER until the
// device LBVH builder lands: every entry point fails loudly.
#include "rtk_dev.h"

#include <stdlib.h>

extern "C" rtk_build *rtk_start_build(const rtk_scene_desc *, rtk_task *) { rtk_set_error("rtk_start_build: device builder not built yet"); return nullptr; }
extern "C" size_t rtk_run_task(const rtk_task *, rtk_task *, size_t) { return 0; }
extern "C" size_t rtk_get_build_size(const rtk_build *) { return 0; }
extern "C" rtk_scene *rtk_finish_build_to(rtk_build *, void *, size_t) { return nullptr; }
extern "C" rtk_scene *rtk_finish_build(rtk_build *) { return nullptr; }
extern "C" rtk_scene *rtk_build_scene(const rtk_scene_desc *) { rtk_set_error("rtk_build_scene: device builder not built yet"); return nullptr; }
extern "C" void rtk_free_scene(rtk_scene *scene) { if (scene) { rtk_amd_forget_scene(scene); free(scene); } }
extern "C" rtk_dev_scene *rtk_dev_scene_build(const rtk_scene_desc *) { rtk_set_error("rtk_dev_scene_build: device builder not built yet"); return nullptr; }
extern "C" size_t rtk_dev_scene_export_size(const rtk_dev_scene *) { return 0; }
extern "C" rtk_scene *rtk_dev_scene_export(const rtk_dev_scene *, void *, size_t) { rtk_set_error("rtk_dev_scene_export: not built yet"); return nullptr; }
