"""libgiddy_amd — MI355X-native lightweight-decompression kernels behind the kernel-wrapper /
launch-configuration / factory API of eyalroz/libgiddy."""
