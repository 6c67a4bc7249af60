// proverServer <port> <circuit1.zkey> ... <circuitN.zkey>
// The REST shell of the reference (src/main_proofserver.cpp:11-45, src/proverapi.cpp:9-41) over a
// self-contained single-threaded HTTP/1.1 loop (the reference's Pistache is an empty submodule):
//   GET  /status            -> FullProver::getStatus() document, application/json
//   POST /input/:circuit    -> body = circom input JSON; 200 at once, job runs in the background
//   POST /cancel            -> abort()
//   POST /start, /stop      -> 200, no-ops (proverapi.cpp:27-33)
// Throughput mode (ZKHIP_QUEUE=n, see fullprover.hpp): POST /input/:circuit answers {"job":id} (503 when n
// requests are already waiting) and GET /status/<id> reports that job; everything else is unchanged.
// One HTTP thread, bodies up to 128000000 bytes (main_proofserver.cpp:32).
#include <arpa/inet.h>
#include <cerrno>
#include <csignal>
#include <cstring>
#include <iostream>
#include <netinet/in.h>
#include <string>
#include <sys/socket.h>
#include <unistd.h>

#include "fullprover.hpp"

static const size_t kMaxRequest = 128000000;

static bool send_all(int fd, const std::string &s) {
    size_t off = 0;
    while (off < s.size()) {
        ssize_t k = ::send(fd, s.data() + off, s.size() - off, MSG_NOSIGNAL);
        if (k <= 0) return false;
        off += (size_t)k;
    }
    return true;
}

static void respond(int fd, int code, const char *reason, const std::string &body, const char *ctype) {
    std::string h = "HTTP/1.1 " + std::to_string(code) + " " + reason + "\r\n";
    if (ctype) h += std::string("Content-Type: ") + ctype + "\r\n";
    h += "Content-Length: " + std::to_string(body.size()) + "\r\nConnection: close\r\n\r\n";
    send_all(fd, h + body);
}

static std::string lower(std::string s) {
    for (auto &c : s) c = (char)tolower((unsigned char)c);
    return s;
}

static void handle(int fd, FullProver &fp) {
    std::string buf;
    size_t hdr_end = std::string::npos;
    char tmp[65536];
    while (hdr_end == std::string::npos) {
        ssize_t k = ::recv(fd, tmp, sizeof tmp, 0);
        if (k <= 0) return;
        buf.append(tmp, (size_t)k);
        hdr_end = buf.find("\r\n\r\n");
        if (hdr_end == std::string::npos && buf.size() > 65536) return respond(fd, 431, "Request Header Fields Too Large", "", nullptr);
    }
    std::string head = buf.substr(0, hdr_end);
    size_t le = head.find("\r\n");
    std::string reqline = head.substr(0, le);
    size_t s1 = reqline.find(' '), s2 = reqline.rfind(' ');
    if (s1 == std::string::npos || s2 == s1) return respond(fd, 400, "Bad Request", "", nullptr);
    std::string method = reqline.substr(0, s1), target = reqline.substr(s1 + 1, s2 - s1 - 1);
    size_t qm = target.find('?');
    if (qm != std::string::npos) target.resize(qm);

    size_t clen = 0;
    bool expect100 = false;
    size_t pos = le == std::string::npos ? head.size() : le + 2;
    while (pos < head.size()) {
        size_t e = head.find("\r\n", pos);
        if (e == std::string::npos) e = head.size();
        std::string line = head.substr(pos, e - pos);
        size_t c = line.find(':');
        if (c != std::string::npos) {
            std::string key = lower(line.substr(0, c)), val = line.substr(c + 1);
            while (!val.empty() && val[0] == ' ') val.erase(0, 1);
            if (key == "content-length") clen = (size_t)strtoull(val.c_str(), nullptr, 10);
            if (key == "expect" && lower(val) == "100-continue") expect100 = true;
        }
        pos = e + 2;
    }
    if (clen > kMaxRequest) return respond(fd, 413, "Request Entity Too Large", "", nullptr);
    std::string body = buf.substr(hdr_end + 4);
    if (expect100 && body.size() < clen) send_all(fd, "HTTP/1.1 100 Continue\r\n\r\n");
    while (body.size() < clen) {
        ssize_t k = ::recv(fd, tmp, sizeof tmp, 0);
        if (k <= 0) return;
        body.append(tmp, (size_t)k);
    }
    body.resize(clen);

    if (method == "GET" && target == "/status") return respond(fd, 200, "OK", fp.getStatus(), "application/json");
    if (method == "GET" && fp.queueMode() && target.rfind("/status/", 0) == 0 && target.size() > 8) {   // throughput mode: one job's document
        char *end = nullptr;
        const unsigned long long id = strtoull(target.c_str() + 8, &end, 10);
        if (*end) return respond(fd, 404, "Not Found", "Could not find a matching route", "text/plain");
        return respond(fd, 200, "OK", fp.getStatus(id), "application/json");
    }
    if (method == "POST" && (target == "/start" || target == "/stop")) return respond(fd, 200, "OK", "", nullptr);
    if (method == "POST" && target == "/cancel") {
        fp.abort();
        return respond(fd, 200, "OK", "", nullptr);
    }
    if (method == "POST" && target.rfind("/input/", 0) == 0 && target.size() > 7 && target.find('/', 7) == std::string::npos) {
        if (fp.queueMode()) {      // ZKHIP_QUEUE=n: requests queue up instead of replacing each other
            uint64_t id = 0;
            if (!fp.enqueue(body, target.substr(7), id)) return respond(fd, 503, "Service Unavailable", "{\"error\":\"queue full\"}", "application/json");
            return respond(fd, 200, "OK", "{\"job\":" + std::to_string(id) + "}", "application/json");
        }
        fp.startProve(body, target.substr(7));
        return respond(fd, 200, "OK", "", nullptr);
    }
    respond(fd, 404, "Not Found", "Could not find a matching route", "text/plain");
}

int main(int argc, char **argv) {
    if (argc < 3) {
        std::cerr << "Invalid number of parameters:\n";
        std::cerr << "Usage: proverServer <port> <circuit1.zkey> <circuit2.zkey> ... <circuitN.zkey> \n";
        return -1;
    }
    try {
        std::cerr << "Initializing server...\n";
        int port = std::stoi(argv[1]);
        std::string *zkeyFileNames = new std::string[argc - 2];
        for (int i = 0; i < argc - 2; i++) zkeyFileNames[i] = argv[i + 2];
        FullProver fullProver(zkeyFileNames, argc - 2);
        delete[] zkeyFileNames;

        signal(SIGPIPE, SIG_IGN);
        int ls = ::socket(AF_INET, SOCK_STREAM, 0);
        if (ls < 0) throw std::runtime_error(std::string("socket: ") + strerror(errno));
        int one = 1;
        setsockopt(ls, SOL_SOCKET, SO_REUSEADDR, &one, sizeof one);
        sockaddr_in addr{};
        addr.sin_family = AF_INET;
        addr.sin_addr.s_addr = htonl(INADDR_ANY);
        addr.sin_port = htons((uint16_t)port);
        if (::bind(ls, (sockaddr *)&addr, sizeof addr) < 0) throw std::runtime_error(std::string("bind: ") + strerror(errno));
        if (::listen(ls, 64) < 0) throw std::runtime_error(std::string("listen: ") + strerror(errno));
        std::cerr << "Server ready on port " << port << "...\n";
        for (;;) {   // one HTTP thread, like Http::Endpoint::options().threads(1)
            int fd = ::accept(ls, nullptr, nullptr);
            if (fd < 0) {
                if (errno == EINTR) continue;
                throw std::runtime_error(std::string("accept: ") + strerror(errno));
            }
            handle(fd, fullProver);
            ::close(fd);
        }
    } catch (std::exception &e) {
        std::cerr << e.what() << '\n';
        return -1;
    }
}
