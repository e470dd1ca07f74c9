#define HTS_VERSION_TEXT "1.23.1-oracle"
