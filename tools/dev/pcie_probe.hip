// tools/dev/pcie_probe.hip -- how fast do host buffers of a drop-in caller reach the GPU: pageable hipMemcpy, hipHostRegister cost,
// pinned async copies, both directions at once (two streams / two threads).   hipcc -O2 -o pcie_probe pcie_probe.hip -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
	const size_t N = 51220480, M = 34172980;
	char* h_in = (char*)malloc(N); char* h_out = (char*)malloc(M);
	memset(h_in, 1, N); memset(h_out, 2, M);
	char *d_in, *d_out; hipMalloc(&d_in, N); hipMalloc(&d_out, M);
	hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
	for (int rep = 0; rep < 3; ++rep) {
		double t = now(); hipMemcpy(d_in, h_in, N, hipMemcpyHostToDevice); double a = now() - t;
		t = now(); hipMemcpy(h_out, d_out, M, hipMemcpyDeviceToHost); double b = now() - t;
		printf("pageable sync: H2D %.3f ms (%.1f GB/s)  D2H %.3f ms (%.1f GB/s)\n", a * 1e3, N / a / 1e9, b * 1e3, M / b / 1e9);
	}
	for (int rep = 0; rep < 2; ++rep) {   // both directions at once from two threads, pageable
		double t = now();
		std::thread th([&] { hipMemcpy(h_out, d_out, M, hipMemcpyDeviceToHost); });
		hipMemcpy(d_in, h_in, N, hipMemcpyHostToDevice); th.join();
		double a = now() - t; printf("pageable, two threads, both directions: %.3f ms\n", a * 1e3);
	}
	for (int rep = 0; rep < 2; ++rep) {   // sliced pageable async on two streams
		double t = now(); const size_t S = 4 << 20;
		for (size_t o = 0; o < N; o += S) { hipMemcpyAsync(d_in + o, h_in + o, (N - o < S) ? N - o : S, hipMemcpyHostToDevice, s1); if (o < M) hipMemcpyAsync(h_out + o, d_out + o, (M - o < S) ? M - o : S, hipMemcpyDeviceToHost, s2); }
		hipStreamSynchronize(s1); hipStreamSynchronize(s2);
		double a = now() - t; printf("pageable async slices on two streams, both directions: %.3f ms\n", a * 1e3);
	}
	for (int rep = 0; rep < 3; ++rep) {
		double t = now(); hipHostRegister(h_in, N, hipHostRegisterDefault); hipHostRegister(h_out, M, hipHostRegisterDefault); double r = now() - t;
		t = now(); hipMemcpyAsync(d_in, h_in, N, hipMemcpyHostToDevice, s1); hipStreamSynchronize(s1); double a = now() - t;
		t = now(); hipMemcpyAsync(h_out, d_out, M, hipMemcpyDeviceToHost, s2); hipStreamSynchronize(s2); double b = now() - t;
		t = now(); hipMemcpyAsync(d_in, h_in, N, hipMemcpyHostToDevice, s1); hipMemcpyAsync(h_out, d_out, M, hipMemcpyDeviceToHost, s2); hipStreamSynchronize(s1); hipStreamSynchronize(s2); double c = now() - t;
		t = now(); hipHostUnregister(h_in); hipHostUnregister(h_out); double ur = now() - t;
		printf("register %.3f ms, unregister %.3f ms; pinned H2D %.3f ms (%.1f GB/s), D2H %.3f ms (%.1f GB/s), both at once %.3f ms\n", r * 1e3, ur * 1e3, a * 1e3, N / a / 1e9, b * 1e3, M / b / 1e9, c * 1e3);
	}
	char* p_in; hipHostMalloc((void**)&p_in, N, hipHostMallocDefault);
	for (int thr = 1; thr <= 8; thr *= 2) {
		double t = now();
		std::thread th[8];
		for (int i = 0; i < thr; ++i) { th[i] = std::thread([&, i] { size_t a = N / thr * i, b = (i == thr - 1) ? N : N / thr * (i + 1); memcpy(p_in + a, h_in + a, b - a); }); }
		for (int i = 0; i < thr; ++i) { th[i].join(); }
		double a = now() - t; printf("host memcpy pageable -> pinned, %d threads: %.3f ms (%.1f GB/s)\n", thr, a * 1e3, N / a / 1e9);
	}
	return 0;
}
