// HTTP/1.1 front end of proverServer: N workers, each poll()s the listening socket and the connections IT accepted.
// (The reference runs Pistache with ONE thread and one request per connection, src/main_proofserver.cpp:29-41; Pistache is
// an empty submodule in the reference checkout.)
//
// Every socket is NON-BLOCKING and every connection carries its own parse state, so a worker never waits for one client:
// a slow or large upload (bodies go up to 128 MB), a client trickling a byte every few seconds or a peer that does not
// read its answer costs that connection's poll slot, nothing else — the cheap /status polls and accept() on the same
// worker go on.  What a worker DOES wait for is the handler itself, which runs on the worker's thread: a /status poll takes
// microseconds, a /witness body is parsed there (tens of milliseconds for a 128 MB image) and the worker's other connections
// wait that long.  Memory is bounded per connection and per worker: a connection whose unsent answers exceed kMaxPendingOut is
// neither read nor parsed until its peer has taken them (a client that pipelines requests and never reads costs 1 MB, not the
// server's memory); a worker holds at most kMaxConns connections (the listening socket is left to the other workers and the
// backlog beyond that) and buffers at most kMaxBigBodies bodies above 1 MB at a time (the next one is answered 503).  Deadlines: a request must be complete `kRequestDeadlineMs` after its first byte, an idle kept-alive
// connection is dropped after `kIdleMs`, a refused request's unread body is drained for at most `kDrainMs` (closing a
// socket with unread data sends a reset that can overtake the answer).  accept() failing with EMFILE & co. pauses
// accepting on that worker for 100 ms instead of spinning on a listening socket that stays readable.
#pragma once
#include <cerrno>
#include <chrono>
#include <cstring>
#include <fcntl.h>
#include <functional>
#include <iostream>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <poll.h>
#include <string>
#include <sys/socket.h>
#include <thread>
#include <unistd.h>
#include <vector>

namespace httpfront {

struct Request {
    std::string method, target, body;
};
struct Response {
    int code = 200;
    const char *reason = "OK";
    std::string body;
    const char *ctype = nullptr;
};
typedef std::function<Response(Request &&)> Handler;      // called on a worker thread; may throw (-> 500)

static const int64_t kRequestDeadlineMs = 120000;  // first byte of a request -> its last byte (a 128 MB body at 1 MB/s and up)
static const int64_t kIdleMs = 30000;              // kept-alive connection with nothing in flight
static const int64_t kDrainMs = 1000;              // reading and dropping what a refused client still sends
static const size_t kMaxHeader = 65536;
static const size_t kMaxPendingOut = (size_t)1 << 20;   // unsent response bytes of one connection beyond which it is not read
static const size_t kMaxConns = 1024;                    // connections one worker holds
static const size_t kBigBody = (size_t)1 << 20, kMaxBigBodies = 4;   // bodies above kBigBody being received by one worker at a time

inline int64_t now_ms() {
    return std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
inline std::string lower(std::string s) {
    for (auto &c : s) c = (char)tolower((unsigned char)c);
    return s;
}

struct Conn {
    int fd = -1;
    std::string in;              // received, not yet consumed
    size_t in_off = 0;           // parse cursor into `in` (non-zero only inside Worker::parse)
    std::string out;             // serialised responses not yet sent
    size_t out_off = 0;
    int64_t last_ms = 0;         // last byte in or out
    int64_t req_start_ms = 0;    // first byte of the request being received (0: between requests)
    // the request whose body is still arriving
    bool have_head = false, keep = false, want_continue = false, big = false;     // big: counted in the worker's big_bodies_
    std::string method, target;
    size_t hdr_len = 0, clen = 0;
    bool close_after_send = false;   // answer, then close
    bool unread = false;             // a request was refused before its body was read: the client may still be sending
    bool peer_closed = false;        // the peer has shut its send side (what it sent before is still answered)
    bool draining = false;           // answer sent, write side shut down: dropping what still arrives
    int64_t drain_until = 0;
    size_t dropped = 0;
};

class Worker {
  public:
    Worker(int listen_fd, size_t max_body, const Handler &h) : ls_(listen_fd), max_body_(max_body), handler_(h) {}

    void run() {
        std::vector<pollfd> pfds;
        for (;;) {
            const int64_t t = now_ms();
            pfds.clear();
            pfds.push_back(pollfd{ls_, (short)(t >= accept_pause_until_ && conns_.size() < kMaxConns ? POLLIN : 0), 0});
            int64_t wake = t + 1000;
            for (auto &c : conns_) {
                short ev = 0;
                // (a connection about to be closed reads nothing more; one whose peer does not take its answers is not read either)
                if (c.draining || (!c.close_after_send && pending(c) <= kMaxPendingOut)) ev |= POLLIN;
                if (c.out_off < c.out.size()) ev |= POLLOUT;
                pfds.push_back(pollfd{c.fd, ev, 0});
                wake = std::min(wake, deadline_of(c));
            }
            if (t < accept_pause_until_) wake = std::min(wake, accept_pause_until_);
            int timeout = (int)std::max<int64_t>(0, wake - t);
            if (conns_.empty() && t >= accept_pause_until_) timeout = -1;
            const int pr = ::poll(pfds.data(), (nfds_t)pfds.size(), timeout);
            if (pr < 0 && errno != EINTR) {
                std::cerr << "poll: " << strerror(errno) << '\n';
                return;
            }
            const int64_t now = now_ms();
            const size_t nconn = conns_.size();          // connections accepted below are polled next turn
            for (size_t i = 0; i < nconn; i++) {
                Conn &c = conns_[i];
                const short re = pr > 0 ? pfds[i + 1].revents : 0;
                bool alive = true;
                if (re & (POLLIN | POLLHUP | POLLERR)) alive = on_readable(c, now);
                if (alive && (re & POLLOUT)) {
                    alive = flush(c, now);
                    if (alive && !c.draining && !c.in.empty()) alive = pump(c, now);      // requests held back by the output bound
                }
                if (alive && now >= deadline_of(c)) alive = false;          // request too slow, idle too long, drain over
                if (!alive) {
                    ::close(c.fd);
                    c.fd = -1;
                    if (c.big) big_bodies_--;
                }
            }
            for (size_t i = 0; i < conns_.size();) {
                if (conns_[i].fd < 0) {
                    if (i + 1 != conns_.size()) conns_[i] = std::move(conns_.back());
                    conns_.pop_back();
                } else {
                    i++;
                }
            }
            if (pr > 0 && (pfds[0].revents & POLLIN)) accept_some(now);
        }
    }

  private:
    int ls_;
    size_t max_body_;
    const Handler &handler_;
    std::vector<Conn> conns_;
    int64_t accept_pause_until_ = 0;
    size_t big_bodies_ = 0;

    static size_t pending(const Conn &c) { return c.out.size() - c.out_off; }
    static int64_t deadline_of(const Conn &c) {
        if (c.draining) return c.drain_until;
        if (c.req_start_ms) return c.req_start_ms + kRequestDeadlineMs;
        return c.last_ms + kIdleMs;          // between requests (or a peer that does not read its answer)
    }

    void accept_some(int64_t now) {
        for (int burst = 0; burst < 16 && conns_.size() < kMaxConns; burst++) {
            int fd = ::accept(ls_, nullptr, nullptr);
            if (fd < 0) {
                if (errno == EAGAIN || errno == EWOULDBLOCK || errno == EINTR || errno == ECONNABORTED) return;   // another worker took it / nothing left
                // EMFILE, ENFILE, ENOBUFS, ENOMEM ...: the listening socket stays readable — do not spin on it
                accept_pause_until_ = now + 100;
                return;
            }
            int fl = fcntl(fd, F_GETFL, 0);
            fcntl(fd, F_SETFL, fl | O_NONBLOCK);
            int on = 1;
            setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &on, sizeof on);
            Conn c;
            c.fd = fd;
            c.last_ms = now;
            conns_.push_back(std::move(c));
        }
    }

    // false: close the connection now
    bool flush(Conn &c, int64_t now) {
        while (c.out_off < c.out.size()) {
            ssize_t k = ::send(c.fd, c.out.data() + c.out_off, c.out.size() - c.out_off, MSG_NOSIGNAL);
            if (k < 0 && errno == EINTR) continue;
            if (k < 0 && (errno == EAGAIN || errno == EWOULDBLOCK)) return true;      // POLLOUT will call again
            if (k <= 0) return false;
            c.out_off += (size_t)k;
            c.last_ms = now;
        }
        c.out.clear();
        c.out_off = 0;
        if (c.close_after_send && !c.draining) {
            if (!c.unread && c.in.empty()) return false;            // nothing unread: plain close
            // a refused request whose body may still be on its way: closing a socket with unread data sends a reset that can
            // overtake the answer — shut the send side, then read and drop for a moment
            ::shutdown(c.fd, SHUT_WR);
            c.draining = true;
            c.drain_until = now + kDrainMs;
            c.in.clear();
        }
        return true;
    }

    void queue_response(Conn &c, const Response &r, bool keep) {
        std::string h = "HTTP/1.1 " + std::to_string(r.code) + " " + r.reason + "\r\n";
        if (r.ctype) h += std::string("Content-Type: ") + r.ctype + "\r\n";
        h += "Content-Length: " + std::to_string(r.body.size()) + (keep ? "\r\nConnection: keep-alive\r\n\r\n" : "\r\nConnection: close\r\n\r\n");
        c.out += h;
        c.out += r.body;
        if (!keep) c.close_after_send = true;
    }
    // malformed / oversized / unsupported: answered before the body was read; the connection ends
    void refuse(Conn &c, int code, const char *reason, const char *text) {
        Response r;
        r.code = code;
        r.reason = reason;
        if (text) {
            r.body = text;
            r.ctype = "text/plain";
        }
        queue_response(c, r, false);
        c.have_head = false;
        c.req_start_ms = 0;
        c.unread = true;
    }

    // false: close now
    bool on_readable(Conn &c, int64_t now) {
        char tmp[65536];
        size_t budget = (size_t)4 << 20;           // per turn and connection: the others get their turn too
        for (;;) {
            ssize_t k = ::recv(c.fd, tmp, sizeof tmp, 0);
            if (k < 0 && errno == EINTR) continue;
            if (k < 0 && (errno == EAGAIN || errno == EWOULDBLOCK)) break;
            if (k < 0) return false;
            if (k == 0) {                          // the peer is done sending; requests it sent before are still answered
                if (c.draining) return false;
                c.peer_closed = true;
                break;
            }
            c.last_ms = now;
            if (c.draining) {
                c.dropped += (size_t)k;
                if (c.dropped > ((size_t)8 << 20)) return false;
                continue;
            }
            if (c.close_after_send) continue;      // (not polled for input; defensive)
            if (!c.req_start_ms) c.req_start_ms = now;
            c.in.append(tmp, (size_t)k);
            if (budget <= (size_t)k) break;
            budget -= (size_t)k;
        }
        if (c.draining) return true;
        return pump(c, now);
    }

    // Answer what c.in holds and send what can be sent, until the input has no complete request left or the peer is not
    // taking its answers (kMaxPendingOut): a fast reader gets all its pipelined requests answered in this turn — nothing but
    // a socket event would call parse() again, and none comes once the kernel's buffers are empty.  false: close now.
    bool pump(Conn &c, int64_t now) {
        bool blocked = false;
        for (;;) {
            const size_t before = c.in.size();
            parse(c);
            if (!flush(c, now)) return false;
            if (c.draining || c.close_after_send) return true;
            blocked = pending(c) > kMaxPendingOut;
            if (blocked || c.in.empty() || c.in.size() == before) break;
        }
        if (c.peer_closed && !blocked) {                       // the peer is done sending and every complete request is answered
            if (pending(c) == 0) return false;                 // nothing left to say
            c.close_after_send = true;                         // (an incomplete request of a peer that is gone is dropped with the connection)
        }
        return true;
    }

    // consume every complete request in c.in.  Requests are taken from a cursor (c.in_off) and the consumed prefix is erased
    // ONCE on the way out: erasing per request made a 4 MB burst of pipelined /status polls cost 120 000 memmoves of 4 MB.
    void parse(Conn &c) {
        parse_requests(c);
        if (c.in_off) {
            c.in.erase(0, c.in_off);
            c.in_off = 0;
        }
    }
    void parse_requests(Conn &c) {
        while (!c.close_after_send && pending(c) <= kMaxPendingOut) {
            if (!c.have_head) {
                const size_t he = c.in.find("\r\n\r\n", c.in_off);
                if (he == std::string::npos) {
                    if (c.in.size() - c.in_off > kMaxHeader) refuse(c, 431, "Request Header Fields Too Large", nullptr);
                    return;
                }
                if (!parse_head(c, he)) return;
            }
            const size_t have = c.in.size() - c.in_off - c.hdr_len;
            if (have < c.clen) {
                if (!c.want_continue) return;
                c.out += "HTTP/1.1 100 Continue\r\n\r\n";
                c.want_continue = false;
                return;
            }
            Request rq;
            rq.method = std::move(c.method);
            rq.target = std::move(c.target);
            if (c.in_off == 0 && c.in.size() == c.hdr_len + c.clen) {           // the usual case: nothing before or behind the body — no second copy of a large body
                c.in.erase(0, c.hdr_len);
                rq.body = std::move(c.in);
                c.in.clear();
            } else {
                rq.body = c.in.substr(c.in_off + c.hdr_len, c.clen);
                c.in_off += c.hdr_len + c.clen;
            }
            c.have_head = false;
            if (c.big) {
                c.big = false;
                big_bodies_--;
            }
            c.req_start_ms = c.in.size() == c.in_off ? 0 : now_ms();
            Response rs;
            try {
                rs = handler_(std::move(rq));
            } catch (std::exception &e) {          // (out of memory for a body, a failing file write): this request fails, the server stays
                rs = Response();
                rs.code = 500;
                rs.reason = "Internal Server Error";
                rs.body = e.what();
                rs.ctype = "text/plain";
                queue_response(c, rs, false);
                return;
            }
            queue_response(c, rs, c.keep);
        }
    }

    // request line + headers of the request at the cursor of c.in (its header ends at absolute position he); false: refused
    bool parse_head(Conn &c, size_t he) {
        const std::string head = c.in.substr(c.in_off, he - c.in_off);
        const size_t le = head.find("\r\n");
        const std::string reqline = head.substr(0, le);
        const size_t s1 = reqline.find(' '), s2 = reqline.rfind(' ');
        if (s1 == std::string::npos || s2 == s1) {
            refuse(c, 400, "Bad Request", nullptr);
            return false;
        }
        c.method = reqline.substr(0, s1);
        c.target = reqline.substr(s1 + 1, s2 - s1 - 1);
        const size_t qm = c.target.find('?');
        if (qm != std::string::npos) c.target.resize(qm);
        const bool http11 = reqline.size() >= 8 && reqline.compare(reqline.size() - 8, 8, "HTTP/1.1") == 0;
        c.keep = http11;                   // HTTP/1.1: persistent unless the client says close; 1.0: the other way round
        c.clen = 0;
        bool expect100 = false, chunked = false;
        size_t pos = le == std::string::npos ? head.size() : le + 2;
        while (pos < head.size()) {
            size_t e = head.find("\r\n", pos);
            if (e == std::string::npos) e = head.size();
            const std::string line = head.substr(pos, e - pos);
            const size_t col = line.find(':');
            if (col != std::string::npos) {
                const std::string key = lower(line.substr(0, col));
                std::string val = line.substr(col + 1);
                while (!val.empty() && val[0] == ' ') val.erase(0, 1);
                if (key == "content-length") c.clen = (size_t)strtoull(val.c_str(), nullptr, 10);
                if (key == "expect" && lower(val) == "100-continue") expect100 = true;
                if (key == "transfer-encoding" && lower(val) != "identity") chunked = true;
                if (key == "connection") c.keep = lower(val) == "keep-alive" ? true : (lower(val) == "close" ? false : c.keep);
            }
            pos = e + 2;
        }
        if (chunked) {                     // (closes: the body cannot be skipped)
            refuse(c, 501, "Not Implemented", "Transfer-Encoding is not supported: send Content-Length");
            return false;
        }
        if (c.clen > max_body_) {
            refuse(c, 413, "Request Entity Too Large", nullptr);
            return false;
        }
        if (c.clen > kBigBody) {
            if (big_bodies_ >= kMaxBigBodies) {          // this worker already buffers its share of large uploads
                refuse(c, 503, "Service Unavailable", "too many large uploads in progress: retry");
                return false;
            }
            c.big = true;
            big_bodies_++;
        }
        c.hdr_len = he + 4 - c.in_off;
        c.have_head = true;
        c.want_continue = expect100;
        return true;
    }
};

// Serves forever on `listen_fd` (bound, listening) with `nthreads` workers; returns when a worker's poll fails.
inline void serve(int listen_fd, size_t nthreads, size_t max_body, const Handler &h) {
    int fl = fcntl(listen_fd, F_GETFL, 0);
    fcntl(listen_fd, F_SETFL, fl | O_NONBLOCK);          // a connection another worker took first: EAGAIN, not a blocked thread
    std::vector<std::thread> pool;
    for (size_t i = 1; i < nthreads; i++) pool.emplace_back([&] { Worker(listen_fd, max_body, h).run(); });
    Worker(listen_fd, max_body, h).run();
    for (auto &t : pool) t.join();
}

}   // namespace httpfront
