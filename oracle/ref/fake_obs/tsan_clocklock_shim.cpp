// TEST INFRASTRUCTURE, ThreadSanitizer build of the harness only (oracle/ref/Makefile, libwfref_tsan.so).
// std::recursive_timed_mutex::try_lock_for -- the reference's audio callback, src/source.cpp:1822-1824 -- ends in glibc's
// pthread_mutex_clocklock, which gcc 11's libtsan does not intercept: the tool would never see that lock taken, report its unlock
// as "unlock of an unlocked mutex" and everything capture_audio touches under it as a race.  This definition (bound inside the
// library by -Wl,-Bsymbolic-functions) routes the call through pthread_mutex_timedlock, which libtsan does intercept.
#include <pthread.h>
#include <time.h>

extern "C" int pthread_mutex_clocklock(pthread_mutex_t *m, clockid_t clock, const struct timespec *abstime)
{
    struct timespec now_c, now_r, abs_r;
    clock_gettime(clock, &now_c);
    clock_gettime(CLOCK_REALTIME, &now_r);
    long long left = (long long)(abstime->tv_sec - now_c.tv_sec) * 1000000000ll + (abstime->tv_nsec - now_c.tv_nsec);
    if(left < 0)
        left = 0;
    const long long t = (long long)now_r.tv_sec * 1000000000ll + now_r.tv_nsec + left;
    abs_r.tv_sec = (time_t)(t / 1000000000ll);
    abs_r.tv_nsec = (long)(t % 1000000000ll);
    return pthread_mutex_timedlock(m, &abs_r);
}
