// cp_server_main.cpp -- native code-predictor server over the cp_* C ABI (include/qwen3tts_cp.h).
//
// Stands where the reference's two native servers stand (dual_npu/code_predictor_cpp/
// code_predictor_server.cpp:455-556 around onnxruntime, dual_npu/code_predictor_ggml/code_pred_server.cpp
// around qwen3-tts.cpp): same socket, same bytes -- a new connection per frame carries f32[1024] hidden +
// i32 code_0 (4100 B), the reply is i32[15] (60 B), then the server closes the connection -- same flags where
// they still mean something, same warm-up request.  The arithmetic is the HIP library's (cp_predict: 16
// positions x 5 layers + 15 heads in one hipGraph on the GPU); there is no CPU path.
//
//   qwen3_cp_server --weights <Q3TTSW1 file | dir with code_predictor_weights.npz> [--codec_emb <dir with
//                   codec_embedding.npy>] [--socket /tmp/qwen3_cp.sock] [--temperature 0.1] [--top_k 50]
//                   [--seed 42] [--device 0]
#include <getopt.h>
#include <signal.h>
#include <sys/socket.h>
#include <sys/un.h>
#include <unistd.h>

#include <cerrno>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/qwen3tts_cp.h"

extern "C" int q3_set_device(int device);

namespace {

constexpr int kHidden = 1024;
volatile sig_atomic_t g_running = 1;
int g_server_fd = -1;

void on_signal(int) {
    g_running = 0;
    if (g_server_fd >= 0) shutdown(g_server_fd, SHUT_RDWR);   // wakes the blocking accept
}

bool recv_exact(int fd, void* buf, size_t n) {
    char* p = (char*)buf;
    while (n > 0) {
        const ssize_t r = recv(fd, p, n, 0);
        if (r == 0) return false;
        if (r < 0) {
            if (errno == EINTR) continue;
            return false;
        }
        p += r;
        n -= (size_t)r;
    }
    return true;
}

bool send_exact(int fd, const void* buf, size_t n) {
    const char* p = (const char*)buf;
    while (n > 0) {
        const ssize_t r = send(fd, p, n, MSG_NOSIGNAL);
        if (r < 0) {
            if (errno == EINTR) continue;
            return false;
        }
        p += r;
        n -= (size_t)r;
    }
    return true;
}

void usage(const char* argv0) {
    fprintf(stderr,
            "usage: %s --weights <container | model dir> [--codec_emb <embeddings dir>] [--socket PATH]\n"
            "          [--temperature T] [--top_k K] [--seed S] [--device N] [--threads N (ignored)]\n",
            argv0);
}

}  // namespace

int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    setvbuf(stderr, nullptr, _IONBF, 0);
    std::string weights, codec_emb, socket_path = "/tmp/qwen3_cp.sock";
    float temperature = 0.1f;
    int top_k = 50, device = 0;
    unsigned long long seed = 42;   // the reference seeds mt19937 with 42 (code_predictor_server.cpp:136)
    static struct option opts[] = {{"weights", required_argument, nullptr, 'w'},
                                   {"codec_emb", required_argument, nullptr, 'c'},
                                   {"socket", required_argument, nullptr, 's'},
                                   {"threads", required_argument, nullptr, 't'},
                                   {"temperature", required_argument, nullptr, 'T'},
                                   {"top_k", required_argument, nullptr, 'k'},
                                   {"seed", required_argument, nullptr, 'S'},
                                   {"device", required_argument, nullptr, 'd'},
                                   {"help", no_argument, nullptr, 'h'},
                                   {nullptr, 0, nullptr, 0}};
    int opt;
    while ((opt = getopt_long(argc, argv, "w:c:s:t:T:k:S:d:h", opts, nullptr)) != -1) {
        switch (opt) {
            case 'w': weights = optarg; break;
            case 'c': codec_emb = optarg; break;
            case 's': socket_path = optarg; break;
            case 't': break;   // CPU thread count of the reference: nothing to set here
            case 'T': temperature = (float)atof(optarg); break;
            case 'k': top_k = atoi(optarg); break;
            case 'S': seed = strtoull(optarg, nullptr, 10); break;
            case 'd': device = atoi(optarg); break;
            case 'h': usage(argv[0]); return 0;
            default: usage(argv[0]); return 1;
        }
    }
    if (weights.empty()) {
        usage(argv[0]);
        return 1;
    }
    signal(SIGINT, on_signal);
    signal(SIGTERM, on_signal);
    signal(SIGPIPE, SIG_IGN);
    if (q3_set_device(device) != 0) {
        fprintf(stderr, "cannot select HIP device %d\n", device);
        return 1;
    }
    void* cp = cp_load(weights.c_str(), codec_emb.empty() ? nullptr : codec_emb.c_str(), 1);
    if (!cp) {
        fprintf(stderr, "cp_load(%s) failed\n", weights.c_str());
        return 1;
    }
    printf("Code predictor loaded: %s (temperature %.3f, top_k %d)\n", weights.c_str(), temperature, top_k);
    {   // the reference servers' warm-up: hidden = 0.1, code_0 = 100
        std::vector<float> h(kHidden, 0.1f);
        int32_t codes[Q3CP_NUM_GROUPS];
        if (cp_predict(cp, h.data(), 100, temperature, top_k, seed, codes) != 0) {
            fprintf(stderr, "warm-up cp_predict failed\n");
            cp_free(cp);
            return 1;
        }
        printf("  warmup result: [");
        for (int i = 0; i < Q3CP_NUM_GROUPS; i++) printf("%d%s", codes[i], i + 1 < Q3CP_NUM_GROUPS ? "," : "");
        printf("]\n");
    }
    unlink(socket_path.c_str());
    g_server_fd = socket(AF_UNIX, SOCK_STREAM, 0);
    sockaddr_un addr;
    memset(&addr, 0, sizeof(addr));
    addr.sun_family = AF_UNIX;
    if (socket_path.size() >= sizeof(addr.sun_path)) {
        fprintf(stderr, "socket path too long\n");
        cp_free(cp);
        return 1;
    }
    strncpy(addr.sun_path, socket_path.c_str(), sizeof(addr.sun_path) - 1);
    if (g_server_fd < 0 || bind(g_server_fd, (sockaddr*)&addr, sizeof(addr)) != 0 || listen(g_server_fd, 5) != 0) {
        perror("socket/bind/listen");
        cp_free(cp);
        return 1;
    }
    printf("Listening on %s\n", socket_path.c_str());
    unsigned long long n_req = 0;
    double total_ms = 0.0;
    while (g_running) {
        const int fd = accept(g_server_fd, nullptr, nullptr);
        if (fd < 0) {
            if (g_running && errno == EINTR) continue;
            break;
        }
        float hidden[kHidden];
        int32_t code_0 = 0;
        if (recv_exact(fd, hidden, sizeof(hidden)) && recv_exact(fd, &code_0, sizeof(code_0))) {
            const auto t0 = std::chrono::steady_clock::now();
            int32_t codes[Q3CP_NUM_GROUPS];
            // a fresh draw stream per request, like the reference's generator advancing from its seed
            if (cp_predict(cp, hidden, code_0, temperature, top_k, seed + ++n_req, codes) == 0) {
                send_exact(fd, codes, sizeof(codes));
                total_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                if (n_req % 100 == 0) printf("  %llu frames, %.3f ms/frame\n", n_req, total_ms / (double)n_req);
            } else {
                fprintf(stderr, "cp_predict failed\n");   // the connection closes without a reply; the client sees a short read
            }
        }
        close(fd);
    }
    if (g_server_fd >= 0) close(g_server_fd);
    unlink(socket_path.c_str());
    cp_free(cp);
    printf("Server stopped.\n");
    return 0;
}
