// plugin_host.cpp — a stand-in for rfw::system's side of the plugin boundary (RFW/system/src/rfw/system.cpp:65-178,
// 247-433): dlopen "HipRT.so" from the given directory, resolve createRenderContext / destroyRenderContext, drive
// the RenderContext through its VIRTUAL interface in the order system::synchronize uses, render, and print a few
// numbers the pytest wrapper checks against the oracle.  Compiled against the restated interface header.
#include "rfw/restated_context.h"

#include <cmath>
#include <cstdio>
#include <dlfcn.h>
#include <string>

typedef rfw::RenderContext *(*CreateFn)();
typedef void (*DestroyFn)(rfw::RenderContext *);
typedef int (*ReadFn)(rfw::RenderContext *, float *);

int main(int argc, char **argv)
{
	const std::string dir = argc > 1 ? argv[1] : ".";
	void *h = dlopen((dir + "/HipRT.so").c_str(), RTLD_NOW);
	if (!h)
	{
		std::fprintf(stderr, "dlopen failed: %s\n", dlerror());
		return 2;
	}
	auto create = (CreateFn)dlsym(h, "createRenderContext");
	auto destroy = (DestroyFn)dlsym(h, "destroyRenderContext");
	auto readfb = (ReadFn)dlsym(h, "hiprtReadFramebuffer");
	if (!create || !destroy || !readfb)
		return 3;
	int rc = 0;
	try
	{
		rfw::RenderContext *ctx = create();
		const uint W = 64, H = 48;
		GLuint tex = 0;
		ctx->init(&tex, W, H);
		ctx->set_setting(rfw::RenderSetting("integrator", "parity"));
		ctx->set_setting(rfw::RenderSetting("jitter", "center"));
		std::vector<glm::vec3> sky(8 * 4, glm::vec3{0.25f, 0.5f, 0.75f});
		ctx->set_sky(sky, 8, 4);
		ctx->set_textures({});
		rfw::DeviceMaterial mat;
		std::memset(&mat, 0, sizeof(mat));
		mat.diffuse[0] = mat.diffuse[1] = mat.diffuse[2] = 0x3800; // 0.5 in binary16
		ctx->set_materials({mat}, {rfw::MaterialTexIds()});
		// one quad facing the camera at z = 4
		const float verts[4][4] = {{-1, -1, 4, 1}, {1, -1, 4, 1}, {1, 1, 4, 1}, {-1, 1, 4, 1}};
		const unsigned idx[2][3] = {{0, 2, 1}, {0, 3, 2}};
		rfw::Triangle tris[2];
		std::memset(tris, 0, sizeof(tris));
		for (auto &t : tris)
		{
			t.lightTriIdx = -1, t.material = 0;
			t.vN0[2] = t.vN1[2] = t.vN2[2] = t.Nz = -1.0f;
		}
		rfw::Mesh mesh;
		mesh.vertices = &verts[0][0], mesh.normals = nullptr, mesh.texCoords = nullptr, mesh.triangles = tris;
		mesh.indices = &idx[0][0], mesh.vertexCount = 4, mesh.triangleCount = 2;
		ctx->set_mesh(0, mesh);
		glm::mat4 M = {{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}};
		glm::mat3 N = {{1, 0, 0, 0, 1, 0, 0, 0, 1}};
		ctx->set_instance(0, 0, M, N);
		rfw::DevicePointLight pl;
		std::memset(&pl, 0, sizeof(pl));
		pl.position[2] = 0.0f, pl.radiance[0] = pl.radiance[1] = pl.radiance[2] = 8.0f, pl.energy = std::sqrt(192.0f);
		rfw::LightCount lc = {0, 1, 0, 0};
		ctx->set_lights(lc, nullptr, &pl, nullptr, nullptr);
		ctx->update();
		ctx->set_probe_index(glm::uvec2{W / 2, H / 2});
		rfw::Camera cam;
		std::memset(&cam, 0, sizeof(cam));
		cam.direction.z = 1.0f, cam.focalDistance = 5.0f, cam.FOV = 40.0f, cam.aspectRatio = float(W) / H, cam.clampValue = 10.0f;
		cam.pixelCount = glm::ivec2{int(W), int(H)};
		ctx->render_frame(cam, rfw::Reset);
		unsigned inst = 99, prim = 99;
		float dist = 0;
		ctx->get_probe_results(&inst, &prim, &dist);
		std::vector<float> img(size_t(W) * H * 4);
		if (readfb(ctx, img.data()) != 0)
			rc = 4;
		const float *c = &img[(size_t(H / 2) * W + W / 2) * 4];
		const float *corner = &img[0];
		const rfw::RenderStats st = ctx->get_stats();
		std::printf("probe %u %u %.6f\ncenter %.6f %.6f %.6f %.1f\ncorner %.6f %.6f %.6f %.1f\nprimary %u targets %zu\n", inst, prim,
					dist, c[0], c[1], c[2], c[3], corner[0], corner[1], corner[2], corner[3], st.primaryCount,
					ctx->get_supported_targets().size());
		// error path: exceptions must cross the boundary as std::runtime_error
		bool threw = false;
		try
		{
			ctx->set_setting(rfw::RenderSetting("integrator", "bogus"));
		}
		catch (const std::runtime_error &e)
		{
			threw = true;
		}
		std::printf("threw %d\n", threw ? 1 : 0);
		ctx->cleanup(); // system::unload calls cleanup(), then destroy calls it again
		destroy(ctx);
	}
	catch (const std::exception &e)
	{
		std::fprintf(stderr, "exception: %s\n", e.what());
		rc = 5;
	}
	dlclose(h);
	return rc;
}
