"""bcftools_amd -- MI355X-native `bcftools mpileup | bcftools call -m` hot path."""
