"""MI355X-native denoising loop of intellerce/controlanimate."""
