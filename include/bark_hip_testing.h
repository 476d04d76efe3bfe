/* bark_hip_testing.h — hooks for the test-suite only.  NOT part of the drop-in boundary (include/bark_hip.h): nothing on the
 * reference's side binds these, and they are inert unless the process was started with $BARK_TEST_HOOKS set (the variable is
 * read once, when libbarkhip.so is loaded; tests/conftest.py sets it). */
#ifndef BARK_HIP_TESTING_H
#define BARK_HIP_TESTING_H

#ifdef __cplusplus
extern "C" {
#endif

/* The k-th launch from now on whose status the library checks reports hipErrorLaunchFailure, so that the error-return paths —
 * helper streams forked, stream capture open — can be exercised on a healthy device (tests/test_gpu_context.py).  k <= 0
 * switches it off; returns the previous countdown.  Process-wide, hence a test hook: without $BARK_TEST_HOOKS it returns -1 and
 * does nothing, and the library's launch checks do not look at it. */
long bark_debug_fail_launch(long k);

#ifdef __cplusplus
}
#endif
#endif
