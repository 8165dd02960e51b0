"""MI355X-native GloVe training hot path: csrc/ (HIP kernels + C ABI), lib/ (built .so),
trainer/ (host-side mirror of the reference's estimator interface), configs/."""
